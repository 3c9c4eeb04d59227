A="5 1024 120 80 64 998 161"
for cfg in "8 8 8 0" "8 16 4 0" "8 8 4 1" "8 4 8 0"; do set -- $cfg
  echo "FWD UW=$1 NW=$2 UN=$3 DB=$4: $(AMDSPEECH_UW=$1 AMDSPEECH_FWD_NW=$2 AMDSPEECH_FWD_UN=$3 AMDSPEECH_FWD_DB=$4 timeout 300 python tools/quick_bench.py $A 2>&1 | grep -E "^fwd")"
done
echo "FWD TILE=8: $(AMDSPEECH_FWD_TILE=8 timeout 300 python tools/quick_bench.py $A 2>&1 | grep -E "^fwd")"
for cfg in "8 8 1" "4 8 1" "16 8 0" "8 16 0"; do set -- $cfg
  echo "BWD NW=$1 UN=$2 DB=$3: $(AMDSPEECH_BWD_NW=$1 AMDSPEECH_BWD_UN=$2 AMDSPEECH_BWD_DB=$3 timeout 300 python tools/quick_bench.py $A 2>&1 | grep -E "^bwd")"
done
