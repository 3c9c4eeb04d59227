for dbg in 0 1 2 3; do
  echo "FWD dbg=$dbg: $(AMDSPEECH_DBG=$dbg AMDSPEECH_UW=8 AMDSPEECH_FWD_NW=8 AMDSPEECH_FWD_UN=8 AMDSPEECH_FWD_DB=0 timeout 200 python tools/quick_bench.py 2>&1 | grep -E "^fwd")"
done
