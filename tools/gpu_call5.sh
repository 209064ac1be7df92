#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/pmc_kernel.sh panel_m128 128 pc '{"kernel":4,"mt":8,"bm":128,"waves":8,"ksplit":4,"pf":4}' 2>&1 | tail -60
