#!/usr/bin/env python3
"""Stage-2 fine-tuning entry point (reference: Train_Stage2_K.py): mirror loss against a frozen Stage-1 teacher,
occlusion masks, 2B flip batch (fal_net_amd.train.stage2_step, Train_Stage2_K.py:233-331).

Same flags as Train_Stage1_K.py (the reference's two scripts share them, Train_Stage2_K.py:30-71) plus `-mirror_loss` and
`--fix_model`, with the reference's Stage-2 defaults (batch 4, lr 5e-5, milestones 5 / 10, 20 epochs, a_sm 0.4 * 2 / 512);
data loading, GPU augmentation, validate() and checkpoints are Train_Stage1_K.main's.
"""
import Train_Stage1_K as base

if __name__ == '__main__':
    base.parser.description = 'FAL_net Stage 2 on MI355X'
    base.parser.add_argument('-mirror_loss', '--a_mr', type=float, default=1)
    base.parser.add_argument('--fix_model', default=None, help='Stage-1 checkpoint of the frozen teacher (reference format); required with -d')
    base.parser.add_argument('--allow-seeded-teacher', action='store_true',
                             help='real-data run WITHOUT Stage-1 checkpoints: seeded (untrained) student and teacher -- tests only')
    base.parser.set_defaults(batch_size=4, lr=0.00005, milestones=[5, 10], epochs=20, a_sm=0.4 * 2 / 512)
    base.args = base.parser.parse_args()
    base.main(step='stage2_step')
