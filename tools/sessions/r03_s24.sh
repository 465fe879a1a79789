#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 600 python tools/stress_streams.py 2>&1 | grep -v amdgpu.ids | tail -5
