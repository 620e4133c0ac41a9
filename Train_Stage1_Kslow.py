#!/usr/bin/env python3
"""Two-view Stage-1 training entry point (reference: Train_Stage1_Kslow.py) on the MI355X implementation.

Same flags as Train_Stage1_K.py with the reference's differing defaults (batch 4, Train_Stage1_Kslow.py:49); every
step runs the model on (left | flip(right)) and averages the losses over both views
(fal_net_amd.train.stage1_slow_step, Train_Stage1_Kslow.py:236-284).
"""
import Train_Stage1_K as base

if __name__ == '__main__':
    base.parser.description = 'FAL_net Stage 1, two-view variant, on MI355X'
    base.parser.set_defaults(batch_size=4)
    base.args = base.parser.parse_args()
    base.main(step='stage1_slow_step')
