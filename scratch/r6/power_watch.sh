#!/bin/bash
# sample power / clocks while the step graph replays (is the sustained step at the board's power cap?)
python bench.py --steps 600 --warmup 20 > /tmp/bench_pw.log 2>&1 &
BP=$!
sleep 45
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "sclk\|mclk\|power\|junction\|edge" | tr '\n' ' ' ; echo
  sleep 1.5
done
wait $BP
tail -c 600 /tmp/bench_pw.log
rocm-smi --showmaxpower 2>/dev/null | grep -i power
