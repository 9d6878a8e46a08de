#!/bin/bash
cd $GRAFT_REPO_ROOT; python scripts/dev/logits_out_probe.py 2>&1 | grep "ms/token"
