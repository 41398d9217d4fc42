#!/bin/bash
# Samples GPU clock / power while a command runs:  tools/clock_watch.sh <out-file> <cmd...>
out=$1; shift
( while true; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Power|junction|Temperature \(Sensor junction\)" | tr '\n' ' ' ; echo; sleep 0.2; done ) > "$out" &
wpid=$!
"$@"
kill $wpid
