#!/bin/bash
# Board power / shader clock / temperature sampled beside a command (rocm-smi every 0.5 s):  tools/power_beside.sh out.log -- <command ...>
OUT=$1; shift; shift
( while true; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | awk '/junction/ {t=$NF} /sclk/ {c=$(NF-0)} /Socket Graphics Package Power|Average Graphics Package Power/ {p=$NF} END {print t, c, p}'; sleep 0.5; done ) > $OUT &
SMI=$!
"$@"
RC=$?
kill $SMI 2>/dev/null
exit $RC
