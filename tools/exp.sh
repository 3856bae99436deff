for f in 1 2; do timeout 300 python tools/fast_prof.py StressN:20 410 50 sweep=1 fast_form=$f 2>&1 | head -2 | tr '\n' ' '; echo; done
for f in 1 2; do timeout 300 python tools/fast_prof.py StressN:16 400 50 sweep=1 fast_form=$f 2>&1 | head -2 | tr '\n' ' '; echo; done
