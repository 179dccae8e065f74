#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
timeout 300 python tools/probe_dqn.py 2>&1 | tail -3
