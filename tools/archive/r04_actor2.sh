#!/bin/bash
timeout 900 python -m pytest tests/test_agents.py -x -q -m gpu 2>&1 | tail -2
bash tools/archive/r04_actor_stats.sh
