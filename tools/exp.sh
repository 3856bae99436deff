for k in 1 2; do timeout 300 python tools/fast_prof.py Stress1000 400 300 sweep=1 2>&1 | head -2 | tr '\n' ' '; echo; done
timeout 300 python tools/fast_prof.py StressN:14 300 100 sweep=1 2>&1 | head -2 | tr '\n' ' '; echo
timeout 300 python tools/fast_prof.py StressN:20 410 50 sweep=1 2>&1 | head -2 | tr '\n' ' '; echo
timeout 300 python tools/fast_prof.py StressN:32 430 30 sweep=1 2>&1 | head -2 | tr '\n' ' '; echo
timeout 300 python tools/fast_prof.py StressN:40 470 20 sweep=1 2>&1 | head -2 | tr '\n' ' '; echo
