#!/bin/bash
# runs the commands given as arguments one after the other; stops at the first that aborted, was killed or timed out
# (a GPU that faulted is not given more work in the same call), goes on after ordinary failures (pytest rc 1)
for c in "$@"; do
  echo "=== $c"
  bash -c "$c"
  rc=$?
  echo "=== rc $rc"
  if [ $rc -ge 124 ]; then echo "=== stopping"; exit $rc; fi
done
exit 0
