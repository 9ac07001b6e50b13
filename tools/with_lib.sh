#!/bin/bash
# A/B runs without any switch in the product: run a command with a variant build IN PLACE of the in-tree library.
#   tools/with_lib.sh <variants/libksa_x.so | main> <command ...>
# The shipped package loads prgs-sdr-kspecanal_amd/libksa.so and nothing else; this script copies the variant over it for
# the duration of the command and puts the original back (on the GPU box the tree is a scratch copy anyway).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
lib=$1; shift
P=$R/prgs-sdr-kspecanal_amd
if [ "$lib" = main ]; then exec "$@"; fi
[ -f "$R/$lib" ] || { echo "with_lib: $R/$lib not found" >&2; exit 2; }
[ -f $P/libksa.so.main ] || cp -p $P/libksa.so $P/libksa.so.main
cp "$R/$lib" $P/libksa.so
KSA_VARIANT=$(basename "$lib" .so | sed 's/^libksa_//') "$@"
rc=$?
cp -p $P/libksa.so.main $P/libksa.so
exit $rc
