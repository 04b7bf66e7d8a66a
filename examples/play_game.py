#!/usr/bin/env python3
"""The reference's single-env loop (baseline/DQN/train_DQN.py:76-110) on the in-process MI355X simulator.

    python examples/play_game.py

Only the two marked lines differ from the reference loop: where `Game` comes from and how it is constructed.
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from xroute_env_amd import Game                     # reference: from baseline_utils import Game
from xroute_env_amd.regions import config_regions

game = Game(regions=config_regions(1, 4))           # reference: game = Game()  (+ OpenROAD simulator over ZMQ)
for episode in range(3):
    state, reset_try_time = game.reset()
    done = False
    total = 0.0
    while not done:
        action = min(game.legal_action_set)          # put your agent here: dqn_agent.take_action(state)
        next_state, done, violation, wirelength, via = game.step(action)
        reward = -1
        reward *= violation * 500 + via * 4 + wirelength * 0.5
        total += reward
        state = next_state
    print(f"episode {episode}: reward {total:.1f}, last observation {tuple(state.shape)}")
