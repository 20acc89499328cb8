#!/bin/bash
# round 3, session 16: operand-fill microbenchmark (tools/experiments/fill_rate.hip)
out=gpurun_out/r3s16; mkdir -p $out
hipcc -O3 --offload-arch=gfx950 tools/experiments/fill_rate.hip -o /tmp/fill_rate 2> $out/build.err || { echo build failed; tail $out/build.err; exit 1; }
timeout 300 /tmp/fill_rate > $out/fill_rate.txt 2>&1; echo "rc=$?"
cat $out/fill_rate.txt
