#!/bin/bash
# sample sclk / power while the default bench replays (DVFS evidence): usage scratch/clock_watch.sh
cd $GRAFT_REPO_ROOT
python bench.py --steps 900 --warmup 50 --no-cpu-baseline --no-roofline > gpurun_out/clock_bench.log 2>&1 &
BP=$!
for i in $(seq 120); do
  s=$(rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed -E 's/.*sclk clock level: [^(]*\(([0-9]+)Mhz\).*/sclk \1/; s/.*Power \(W\): ([0-9.]+).*/W \1/' | tr '\n' ' ')
  echo "$i $s"
  kill -0 $BP 2>/dev/null || break
  sleep 0.4
done
wait $BP
grep '^{"metric"' gpurun_out/clock_bench.log | cut -c1-200
