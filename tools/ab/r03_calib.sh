#!/bin/bash
# Round-3 calibration (run via gpurun): the pure-stream issue costs again, in s_memtime ticks AND in real time
# (s_memrealtime, 100 MHz), and what the driver says the shader clock is while a stream runs (rocm-smi sampled
# beside a ~4 s stream) -- to settle what an s_memtime tick is on this GPU.
OUT=/root/repo/gpurun_out/r03_calib
rm -rf $OUT; mkdir -p $OUT
CAL=/root/repo/tools/calib/valu_calib
cd /tmp && export TMPDIR=/tmp
$CAL 100000 > $OUT/calib_plain.txt 2>&1
for kind in 3 0 1; do
  ( for i in $(seq 1 14); do /opt/rocm/bin/rocm-smi --showclocks --json 2>/dev/null | tr -d '\n'; echo; sleep 0.3; done ) > $OUT/smi_kind$kind.txt &
  SMI=$!
  sleep 0.5
  $CAL 2000000 $kind > $OUT/long_kind$kind.txt 2>&1
  wait $SMI
done
/opt/rocm/bin/rocm-smi --showclocks > $OUT/smi_idle.txt 2>&1
cat $OUT/calib_plain.txt $OUT/long_kind*.txt
grep -o '"sclk clock speed:": "[^"]*"' $OUT/smi_kind*.txt | sort | uniq -c
head -c 600 $OUT/smi_kind3.txt
