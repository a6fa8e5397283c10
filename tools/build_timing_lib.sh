#!/bin/bash
# Instrumented copies of libcpc_hip.so for the in-kernel traces (never the product library):
#   tools/_bin/libcpc_timing.so      conv_dma.hip with -DCPC_DMA_TIMING (s_memtime stamps of tools/time_dma_slots.py)
#   tools/_bin/libcpc_gru_timing.so  gru.hip with -DCPC_GRU_TIMING (phase laps of tools/time_gru_phases.py)
set -e
cd "$(dirname "$0")/.."
python -m cpc_audio_amd.build
mkdir -p tools/_bin
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -Wno-unused-result -fno-slp-vectorize"
hipcc $FLAGS -DCPC_DMA_TIMING -c cpc_audio_amd/csrc/conv_dma.hip -o tools/_bin/conv_dma_timing.o
objs=$(ls cpc_audio_amd/lib/obj/*.o | grep -v conv_dma.hip.o)
hipcc -shared -fPIC --offload-arch=gfx950 $objs tools/_bin/conv_dma_timing.o -o tools/_bin/libcpc_timing.so
echo tools/_bin/libcpc_timing.so
hipcc $FLAGS -DCPC_GRU_TIMING -c cpc_audio_amd/csrc/gru.hip -o tools/_bin/gru_timing.o
objs=$(ls cpc_audio_amd/lib/obj/*.o | grep -v gru.hip.o)
hipcc -shared -fPIC --offload-arch=gfx950 $objs tools/_bin/gru_timing.o -o tools/_bin/libcpc_gru_timing.so
echo tools/_bin/libcpc_gru_timing.so
