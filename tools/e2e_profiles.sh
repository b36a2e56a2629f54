#!/bin/bash
# per-kernel tables of the end-to-end fusions (run on the GPU box from the repo root)
set -e
T=$GRAFT_REPO_ROOT/tools
bash $T/e2e_profile_one.sh kf $T/e2e_vgg.py --algo kf --models 4
bash $T/e2e_profile_one.sh gr $T/e2e_vgg.py --algo gr
bash $T/e2e_profile_one.sh otf $T/e2e_vgg.py --algo otf
bash $T/e2e_profile_one.sh alf $T/e2e_alf_vit.py
