for k in 0 96 128 160 200; do timeout 300 python tools/fast_prof.py Stress1000 400 300 sweep=1 fast_quiet=$k 2>&1 | head -2 | tr '\n' ' '; echo; done
for k in 0 128 160; do timeout 300 python tools/fast_prof.py StressN:14 300 100 sweep=1 fast_quiet=$k 2>&1 | head -2 | tr '\n' ' '; echo; done
for k in 0 128 160; do timeout 300 python tools/fast_prof.py StressN:20 410 64 sweep=1 fast_quiet=$k 2>&1 | head -2 | tr '\n' ' '; echo; done
