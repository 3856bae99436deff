for k in 0 1 2 3; do timeout 300 python tools/fast_prof.py Stress1000 400 300 sweep=1 fast_sleep=$k 2>&1 | head -2 | tr '\n' ' '; echo; done
for k in 0 1 2 3; do timeout 300 python tools/fast_prof.py StressN:20 410 50 sweep=1 fast_sleep=$k 2>&1 | head -2 | tr '\n' ' '; echo; done
for k in 0 2; do timeout 300 python tools/fast_prof.py StressN:40 470 20 sweep=1 fast_sleep=$k 2>&1 | head -2 | tr '\n' ' '; echo; done
