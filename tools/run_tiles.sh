for sb in 1 2; do echo "== side_build $sb"; python tools/first_call.py 1000000 4 side_build=$sb 2>&1 | grep -E "first call|counters"; done
