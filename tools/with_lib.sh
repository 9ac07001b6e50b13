#!/bin/bash
# A/B runs without any switch in the product: run a command with a variant build IN PLACE of the in-tree library.
#   tools/with_lib.sh <variants/libksa_x.so | main> <command ...>
# The shipped package loads prgs-sdr-kspecanal_amd/libksa.so and nothing else; this script copies the variant over it for
# the duration of the command and puts the original back -- also when the command is interrupted or killed by a signal
# the shell can catch (trap on EXIT).  Copies keep their timestamps (cp -p), so build.is_stale() still sees a library that
# is older than its sources if an experiments build were ever left behind.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
lib=$1; shift
P=$R/prgs-sdr-kspecanal_amd
if [ "$lib" = main ]; then exec "$@"; fi
[ -f "$R/$lib" ] || { echo "with_lib: $R/$lib not found" >&2; exit 2; }
[ -f $P/libksa.so.main ] || cp -p $P/libksa.so $P/libksa.so.main
restore() { [ -f $P/libksa.so.main ] && cp -p $P/libksa.so.main $P/libksa.so; }
trap restore EXIT
trap 'exit 130' INT TERM HUP
cp -p "$R/$lib" $P/libksa.so
KSA_VARIANT=$(basename "$lib" .so | sed 's/^libksa_//') "$@"
exit $?
