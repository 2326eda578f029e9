#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
GRIT_AB_OUT=gpurun_out/r03/ab2 bash tools/micro/ab_env.sh GRIT_ADAM_NT 0 1
