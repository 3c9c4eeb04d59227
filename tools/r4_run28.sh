mkdir -p gpurun_out/r4y
AMDSPEECH_LIB=$(pwd)/tools/variants/beamtiming.so timeout 900 python tools/pinned_read.py > gpurun_out/r4y/a.log 2>&1
grep -E "beam timing|pinned|pageable" gpurun_out/r4y/a.log | tail -6
AMDSPEECH_LIB=$(pwd)/tools/variants/beamtiming.so timeout 1500 python tools/dropin_decoder_sweep.py > gpurun_out/r4y/b.log 2>&1
grep -E "ms per step|beam timing" gpurun_out/r4y/b.log | tail -14
