#!/bin/bash
# tools/r05_ab.sh lib1 lib2 ... : the four triangle scenes under each device library (NODE_FORMAT from the environment)
for lib in "$@"; do MOPTIX_DEVICE_LIB=$lib timeout 600 python3 tools/scene_times.py 2>&1 | grep -v "^\[moptix\]"; done
