#!/bin/bash
# Run on the GPU box: kernel trace of the overlap leg under each env setting given ("A=1,B=2" per argument).
for s in "$@"; do
  echo "== $s"
  env $(echo $s | tr ',' ' ') tools/trace_overlap.sh
done
