for k in 1 2 4 8; do timeout 300 python tools/fast_prof.py Stress1000 400 300 sweep=1 fast_recolor=$k 2>&1 | head -2 | tr '\n' ' '; echo; done
for k in 1 2 4 8 16; do timeout 300 python tools/fast_prof.py StressN:20 410 64 sweep=1 fast_recolor=$k 2>&1 | head -2 | tr '\n' ' '; echo; done
for k in 4 8 16; do timeout 300 python tools/fast_prof.py StressN:40 470 32 sweep=1 fast_recolor=$k 2>&1 | head -2 | tr '\n' ' '; echo; done
