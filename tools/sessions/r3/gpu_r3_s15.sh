#!/bin/bash
# round 3, session 15: capture after an eager forward (fixed), the graph / backbone / generation tests, one bench line
out=gpurun_out/r3s15; mkdir -p $out
timeout 900 python -m pytest tests/test_hip_graph.py tests/test_hip_backbones.py tests/test_model_plumbing.py tests/test_hip_benchpath.py -q -m gpu -x > $out/pytest.txt 2>&1; echo "pytest rc=$?"
tail -4 $out/pytest.txt | cut -c1-600
timeout 600 python bench.py --steps 10 --warmup 3 --companions off > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python -c "import json;d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['caption']['value'],d['caption']['ms_per_decode_step'])"
