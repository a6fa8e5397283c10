#!/bin/bash
# libcpc_hip.so with the s_memtime stamps of tools/time_dma_slots.py compiled in (conv_dma.hip with -DCPC_DMA_TIMING)
set -e
cd "$(dirname "$0")/.."
python -m cpc_audio_amd.build
mkdir -p tools/_bin
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -Wno-unused-result -DCPC_DMA_TIMING \
      -c cpc_audio_amd/csrc/conv_dma.hip -o tools/_bin/conv_dma_timing.o
objs=$(ls cpc_audio_amd/lib/obj/*.o | grep -v conv_dma.hip.o)
hipcc -shared -fPIC --offload-arch=gfx950 $objs tools/_bin/conv_dma_timing.o -o tools/_bin/libcpc_timing.so
echo tools/_bin/libcpc_timing.so
