for cfg in "8 8 8 0" "8 8 4 1" "8 8 2 1" "8 8 1 1" "8 8 2 0" "8 4 4 1" "8 4 2 1"; do set -- $cfg
  echo "FWD UW=$1 NW=$2 UN=$3 DB=$4: $(AMDSPEECH_UW=$1 AMDSPEECH_FWD_NW=$2 AMDSPEECH_FWD_UN=$3 AMDSPEECH_FWD_DB=$4 timeout 200 python tools/quick_bench.py 2>&1 | grep -E "^fwd")"
done
