#!/bin/bash
# tools/build_variant.sh <name> "<extra cxx flags>": a variant libamdspeech under tools/variants/<name>.so (own object dir)
name=$1; shift
AMDSPEECH_LIB_OUT=$(pwd)/tools/variants/$name.so AMDSPEECH_CXXFLAGS="$*" python rnn-speech_amd/build.py 2>&1 | grep -E "error|warning: v|Error" | head -20
ls -la tools/variants/$name.so | awk '{print $5, $9}'
