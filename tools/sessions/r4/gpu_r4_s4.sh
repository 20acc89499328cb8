#!/bin/bash
# round 4, session 4: decode kernels after the rewrite (preloaded arguments, LayerNorm with independent accumulators, write-through slabs): probe,
# parity (incl. the tiny case with the decode path switched off, to see whose failure it is), the chain, the caption leg
ulimit -c 0
tag=${1:-r4s4}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
P=tools/experiments/_bin/decode_probe
for a in "32 5120 1280 20 1 1" "32 5120 1280 20 1 0" "32 1280 5120 20 4 0" "15 256 256 4 1 1" "32 8192 2048 32 1 1" "32 2048 8192 32 4 0" "7 3072 768 12 1 1" "7 768 3072 12 2 0"; do
  echo "== $a"; timeout 60 $P $a 2>&1 | tail -3
done > $out/probe.txt 2>&1
cat $out/probe.txt
timeout 300 python -m pytest tests/test_hip_modules.py -m gpu -q -p no:cacheprovider -k "decode_shaped or resident" 2>&1 | tail -5
( export FLAMINGO_FUSION_LIB=debug FF_DECODE_FFW=0; timeout 300 python -m pytest tests/test_hip_modules.py -m gpu -q -p no:cacheprovider -k "decode_shaped" 2>&1 | tail -4 )
python tools/decode_chain_bench.py
for v in "FF_DECODE_FFW=0" "FF_DECODE_FFW=1" "FF_DECODE_FFW=0" "FF_DECODE_FFW=1"; do
  ( export FLAMINGO_FUSION_LIB=debug $v; echo "[$v] $(python tools/decode_chain_bench.py 2>&1 | tail -1)" )
done
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 --companions off 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d['caption']; print('caption', c['value'], 'tok/s', c['ms_per_decode_step'], 'ms/step', c['library'])"
