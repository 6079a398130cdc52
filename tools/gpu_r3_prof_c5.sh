#!/bin/bash
set -u; timeout 600 tools/gpu_profile.sh r03 c5
