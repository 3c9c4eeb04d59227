for cfg in "8 8 8 0" "8 4 8 0" "8 4 16 0" "8 16 4 0" "8 8 4 1" "4 8 8 0"; do set -- $cfg
  echo "FWD UW=$1 NW=$2 UN=$3 DB=$4: $(AMDSPEECH_UW=$1 AMDSPEECH_FWD_NW=$2 AMDSPEECH_FWD_UN=$3 AMDSPEECH_FWD_DB=$4 timeout 200 python tools/quick_bench.py 2>&1 | grep -E "^fwd")"
done
for cfg in "4 8 1" "8 8 1" "8 16 0" "16 8 0" "4 16 0"; do set -- $cfg
  echo "BWD NW=$1 UN=$2 DB=$3: $(AMDSPEECH_BWD_NW=$1 AMDSPEECH_BWD_UN=$2 AMDSPEECH_BWD_DB=$3 timeout 200 python tools/quick_bench.py 2>&1 | grep -E "^bwd")"
done
