#!/bin/bash
# Board power / shader clock / temperature sampled beside a command (rocm-smi every 0.5 s):  tools/power_beside.sh out.log -- <command ...>
# The sampler runs in its own process group and the WHOLE group is ended when the command returns or the script is interrupted (a bare
# `kill $!` ends the sub-shell only: an in-flight rocm-smi | awk or the sleep outlives it, and an interrupted script orphans the loop).
OUT=$1; shift; shift
set -m                                   # job control: the background job gets its own process group
( while true; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | awk '/junction/ {t=$NF} /sclk/ {c=$(NF-0)} /Socket Graphics Package Power|Average Graphics Package Power/ {p=$NF} END {print t, c, p}'; sleep 0.5; done ) > "$OUT" &
SMI=$!
stop_sampler() { kill -- -"$SMI" 2>/dev/null; wait "$SMI" 2>/dev/null; }
trap 'stop_sampler; exit 130' INT TERM
trap stop_sampler EXIT
"$@"
RC=$?
exit $RC
