"""Hyper-parameters of the three shipped experiments, as flat attribute dicts.

The reference flattens its two-level YAML files into one EasyDict
(src/utils/misc.py:10-29); only the keys the hot path reads are kept here
("Keys the hot path actually reads").  Values transcribed
from src/conf/qk_regtr_full_{3dmatch,kitti,modelnet}.yaml.  `load_config` also
accepts a path to one of those YAML files and flattens it the same way.
"""
import copy


class Config(dict):
    """dict with attribute access; missing attributes raise AttributeError so
    that copy.deepcopy and hasattr behave."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return Config(copy.deepcopy(dict(self), memo))


_COMMON = dict(
    model='qk_regtr_full.RegTR',
    aggregation_mode='sum', fixed_kernel_points='center', in_feats_dim=1, in_points_dim=3,
    deform_radius=5.0, KP_extent=2.0, KP_influence='linear', use_batch_norm=True,
    batch_norm_momentum=0.02, modulated=False, num_kernel_points=15,
    use_lgr=False, use_ransac=False, remove_points_from_val=False, threshold_corr=False,
    remove_outliers_overlap=False, use_overlap_as_weights=False, use_ratio_test=False,
    lowe_thres=0.9, use_attn_affinity=False, use_corr_affinity=False,
    attention_type='dot_prod', nhead=8, d_embed=256, d_feedforward=1024, dropout=0.0,
    pre_norm=True, transformer_act='relu', num_encoder_layers=6,
    transformer_encoder_has_pos_emb=True, sa_val_has_pos_emb=True, ca_val_has_pos_emb=True,
    pos_emb_type='sine', feature_loss_type='infonce', wt_feature=0.1, wt_feature_un=0.0,
    wt_overlap=1.0, wt_corr=1.0,
    # solver section of the YAML files (read by training.configure_optimizers / Trainer)
    optimizer='AdamW', base_lr=0.0001, weight_decay=0.0001, grad_clip=0.1, scheduler='step',
    scheduler_param=[127800, 0.5], reg_success_thresh_rot=10, reg_success_thresh_trans=0.1,
)

_RESNET = ['resnetb', 'resnetb_strided', 'resnetb', 'resnetb', 'resnetb_strided', 'resnetb', 'resnetb']

CONFIGS = {
    # conf/qk_regtr_full_3dmatch.yaml
    '3dmatch': dict(
        dataset='3dmatch', neighborhood_limits=[40, 40, 40, 40], first_subsampling_dl=0.025,
        first_feats_dim=128, conv_radius=2.5, architecture=['simple'] + _RESNET,
        use_sinkhorn=True, sinkhorn_itr=3, slack=True, r_p=0.2, r_n=0.4, val_threshold=0.15,
        num_refinement_steps=4, acceptance_radius=0.1,
    ),
    # conf/qk_regtr_full_kitti.yaml
    'kitti': dict(
        dataset='kitti', neighborhood_limits=[39, 57, 68, 74], first_subsampling_dl=0.2,
        first_feats_dim=128, conv_radius=4.25,
        architecture=['simple'] + _RESNET + ['resnetb_strided', 'resnetb', 'resnetb'],
        use_sinkhorn=False, sinkhorn_itr=3, slack=True, r_p=1.6, r_n=3.2, val_threshold=0.25,
        num_refinement_steps=10, acceptance_radius=0.6,
        scheduler_param=[135800, 0.5], reg_success_thresh_rot=5, reg_success_thresh_trans=2,
    ),
    # conf/qk_regtr_full_modelnet.yaml
    'modelnet': dict(
        dataset='modelnet', neighborhood_limits=[50, 50], first_subsampling_dl=0.03,
        first_feats_dim=512, conv_radius=2.75,
        architecture=['simple', 'resnetb', 'resnetb', 'resnetb_strided', 'resnetb', 'resnetb'],
        use_sinkhorn=False, sinkhorn_itr=1, slack=False, r_p=0.12, r_n=0.24,
        num_refinement_steps=5, acceptance_radius=0.05,
    ),
}


def get_config(name: str) -> Config:
    cfg = Config(copy.deepcopy(_COMMON))
    cfg.update(copy.deepcopy(CONFIGS[name]))
    return cfg


def load_config(path: str) -> Config:
    """Flatten a reference-style two-level YAML (later sections override)."""
    import yaml
    with open(path) as f:
        raw = yaml.safe_load(f)
    flat = Config()
    for section in raw.values():
        if isinstance(section, dict):
            flat.update(section)
    return flat
