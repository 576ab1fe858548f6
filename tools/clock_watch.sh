#!/bin/bash
# Samples the shader clock while a command runs:  tools/clock_watch.sh OUT -- cmd ...
out=$1; shift; shift
( for i in $(seq 1 400); do /opt/rocm/bin/rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -1; sleep 0.05; done ) > "$out" &
w=$!
"$@"
kill $w 2>/dev/null
