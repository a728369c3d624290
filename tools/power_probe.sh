#!/bin/bash
# Socket power, clocks and temperature while the headline bench runs (rocm-smi sampled every 0.5 s): is the sustained clock the
# power cap's?   bash tools/power_probe.sh <tag> [bench args]  -> gpurun_out/<tag>_power_probe.txt
tag=${1:-power}; shift
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
out=gpurun_out/${tag}_power_probe.txt
{ echo "# idle:"; rocm-smi --showpower --showclocks --showtemp --showmaxpower 2>&1 | grep -v "^=\|^$" | head -30; } > $out
python bench.py --steps 24 --warmup 4 --cpu-sample 0 "$@" > /tmp/pp_bench.json 2>/dev/null &
pid=$!
sleep 8   # import + keys + warm-up
echo "# under load (every 0.5 s):" >> $out
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -i "power\|sclk\|mclk\|fclk\|socclk\|Temperature (Sensor junction)\|hotspot\|edge" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ';' >> $out
  echo >> $out
  sleep 0.5
done
tail -1 /tmp/pp_bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('# bench:', round(d['value']), 'gates/s', d['ms_per_step'], 'ms/step')" >> $out
cat $out | cut -c1-400
