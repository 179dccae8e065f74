#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
python tools/dbg_policy.py 2>&1 | tail -6
