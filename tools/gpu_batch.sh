#!/bin/bash
# GPU batch (round 6): closing suite + the driver's command with the final library (split plan in)
bash tools/gpu_suite.sh r06
bash tools/gpu_evidence.sh r06 bench
bash tools/gpu_evidence.sh r06 update
