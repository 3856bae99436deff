timeout 300 python tools/fast_prof.py Stress1000 400 200 sweep=1 2>&1 | head -2
timeout 300 python tools/fast_prof.py Stress1000 400 200 sweep=1 fast_recolor=1 2>&1 | head -2
timeout 300 python tools/fast_prof.py StressN:20 410 50 sweep=1 2>&1 | head -2
timeout 300 python tools/fast_prof.py StressN:20 410 50 sweep=1 fast_recolor=1 2>&1 | head -2
timeout 300 python tools/fast_prof.py StressN:40 470 20 sweep=1 2>&1 | head -2
timeout 300 python tools/fast_prof.py StressN:40 470 20 sweep=1 fast_recolor=1 2>&1 | head -2
