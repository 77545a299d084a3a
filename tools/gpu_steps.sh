#!/bin/bash
# Run a list of GPU steps in ONE gpurun call: an ordinary failure (rc 1..123: a failing test, a Python exception) is logged and
# the next step runs; a step that was killed, timed out or died on a signal (rc >= 124) stops the call -- nothing further
# touches the GPU.  usage: tools/gpu_steps.sh "name1::cmd1" "name2::cmd2" ...   (logs: gpurun_out/<name>.log)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
mkdir -p gpurun_out
for step in "$@"; do
    name="${step%%::*}"
    cmd="${step#*::}"
    echo "=== $name: $cmd"
    bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
    rc=$?
    echo "=== $name rc=$rc"
    tail -n 6 "gpurun_out/$name.log"
    if [ $rc -ge 124 ]; then
        echo "=== $name ended abnormally (rc=$rc): stopping, no further GPU step in this call"
        exit $rc
    fi
done
exit 0
