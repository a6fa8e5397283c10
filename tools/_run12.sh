#!/bin/bash
export TMPDIR=/tmp
python - <<'PY' 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -4
import sys, pytest
from cpc_audio_amd import _lib
_lib.get().check(_lib.get().cpc_set_bf16_dma_depth(1), "depth")
sys.exit(pytest.main(["tests/test_gpu_bf16.py", "-m", "gpu", "-q", "-x"]))
PY
ROUNDS=2 bash tools/ab_bench.sh gpurun_out/r6_ab_bf16_depth "d0|--dtype bf16" "d1|--dtype bf16 --call cpc_set_bf16_dma_depth=1"
rm -rf /tmp/pb; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pb -o res -- python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-probes --launch eager --sustained-seconds 0 --call cpc_set_bf16_dma_depth=1 > /dev/null 2>&1
db=$(find /tmp/pb -name "*.db" | head -1); python tools/step_timeline.py $db --stats | grep "conv_fwd_dma\|conv_dgrad_dma\|wgrad" 
