#!/bin/bash
# GPU batch (round 6): the driver's command with the final library (second closing sample), update / fit timings
bash tools/gpu_evidence.sh r06 bench
bash tools/gpu_evidence.sh r06 fit
