"""CPU checks of the host-side train-loop pieces: the cosine schedule of tools/create_scheduler.py:20-32 (pinned to the closed form),
the reference's lr_scheduler.step(epoch) call pattern (main.py:434), wrapper detection, and the arch registry."""
import math
import types

import pytest
import torch


class _Opt:
    def __init__(self, lrs):
        self.param_groups = [dict(lr=v, initial_lr=v) for v in lrs]


def _closed_form(t, base, t_initial=200, lr_min=1e-5, warm_init=1e-4, warm_t=5):
    if t < warm_t:
        return warm_init + t * (base - warm_init) / warm_t
    if t < t_initial:
        return lr_min + 0.5 * (base - lr_min) * (1.0 + math.cos(math.pi * t / t_initial))
    return lr_min


def test_cosine_schedule_matches_closed_form_table():
    from protopformer_amd.engine import CosineLRScheduler
    bases = [1e-4, 3e-3, 3e-3, 3e-3]                                   # features / add-on / prototypes / global (train_cub.sh:22-24)
    opt = _Opt(bases)
    s = CosineLRScheduler(opt, t_initial=200, lr_min=1e-5, warmup_lr_init=1e-4, warmup_t=5)
    assert [g["lr"] for g in opt.param_groups] == [1e-4] * 4           # construction sets warmup_lr_init
    table = {0: 1e-4, 1: 1e-4 + (3e-3 - 1e-4) / 5, 4: 1e-4 + 4 * (3e-3 - 1e-4) / 5, 5: 1e-5 + 0.5 * (3e-3 - 1e-5) * (1 + math.cos(math.pi * 5 / 200)),
             100: 1e-5 + 0.5 * (3e-3 - 1e-5), 199: 1e-5 + 0.5 * (3e-3 - 1e-5) * (1 + math.cos(math.pi * 199 / 200)), 200: 1e-5, 209: 1e-5}
    for t in range(0, 212):
        s.step(t)
        for g, b in zip(opt.param_groups, bases):
            assert g["lr"] == pytest.approx(_closed_form(t, b), rel=1e-12, abs=0)
        if t in table:
            assert opt.param_groups[1]["lr"] == pytest.approx(table[t], rel=1e-12)
    # features group: base lr == warm-up lr -> flat through the warm-up, cosine afterwards
    s.step(3)
    assert opt.param_groups[0]["lr"] == pytest.approx(1e-4)
    assert s.get_cycle_length() == 200


def test_scheduler_state_dict_round_trip_and_factory():
    from protopformer_amd.engine import CosineLRScheduler, create_scheduler
    args = types.SimpleNamespace(sched="cosine", epochs=200, min_lr=1e-5, warmup_lr=1e-4, warmup_epochs=5, cooldown_epochs=10)
    opt = _Opt([1e-4, 3e-3])
    s, n_epochs = create_scheduler(args, opt)
    assert n_epochs == 210                                             # create_scheduler.py:33
    s.step(37)
    sd = s.state_dict()
    assert "optimizer" not in sd and sd["t_initial"] == 200
    opt2 = _Opt([1e-4, 3e-3])
    s2 = CosineLRScheduler(opt2, t_initial=7, warmup_t=1)
    s2.load_state_dict(sd)
    s2.step(38); s.step(38)
    assert [g["lr"] for g in opt2.param_groups] == [g["lr"] for g in opt.param_groups]
    with pytest.raises(NotImplementedError):
        create_scheduler(types.SimpleNamespace(sched="step", epochs=5), opt)


def test_data_parallel_wrappers_are_rejected():
    """under DistributedDataParallel every parameter looks unused (gradients go straight into the flat buffer)."""
    from protopformer_amd.engine import train_one_step
    m = torch.nn.DataParallel(torch.nn.Linear(2, 2))
    with pytest.raises(RuntimeError, match="DistributedDataParallel"):
        train_one_step(m, None, None, None, None)


def test_registered_architectures():
    from protopformer_amd.protopformer import ARCHS, build_features
    assert {"deit_tiny_patch16_224", "deit_small_patch16_224", "deit_base_patch16_224", "cait_xxs24_224"} <= set(ARCHS)
    f = build_features("deit_base_patch16_224", pretrained=False)
    assert f.embed_dim == 768 and f.num_heads == 12 and len(f.blocks) == 12
    rates = f.droppath_rates()
    assert rates[0] == 0.0 and rates[-1] == pytest.approx(0.1)          # stochastic depth decay rule (deit:89)


def test_parameter_order_matches_reference_state_dict():
    """torch.optim.AdamW checkpoints index parameters by position: our registration order must equal the reference's."""
    from helpers import micro
    from protopformer_amd.cait import MyCait
    from protopformer_amd.deit import MyVisionTransformer
    from protopformer_amd.protopformer import PPNet
    for name in ("micro_deit.npz", "micro_cait.npz"):
        sd, cfg, z = micro(name)
        if cfg["arch"] == "deit":
            feats = MyVisionTransformer(img_size=64, patch_size=16, embed_dim=cfg["dim"], depth=cfg["depth"], num_heads=cfg["heads"], drop_path_rate=0.0)
        else:
            feats = MyCait(img_size=64, patch_size=16, embed_dim=cfg["dim"], depth=cfg["depth"], num_heads=cfg["heads"], drop_path_rate=0.0)
        m = PPNet(features=feats, img_size=64, prototype_shape=[20, 32, 1, 1], proto_layer_rf_info=None, num_classes=10,
                  reserve_layers=[cfg["reserve_layer"]], reserve_token_nums=[9], use_global=True, use_ppc_loss=True, global_proto_per_class=2,
                  add_on_layers_type="regular")
        ref_keys = [k[3:] for k in z.files if k.startswith("sd/")]
        mine = list(m.state_dict().keys())
        # module-wise order inside `features` and the relative order of the head tensors
        assert [k for k in mine if k.startswith("features.")] == [k for k in ref_keys if k.startswith("features.")], name
        assert [k for k in mine if k.startswith("add_on")] == [k for k in ref_keys if k.startswith("add_on")]


def test_graphed_step_refuses_a_phase_it_was_not_captured_for():
    """the captured graph bakes in the epoch >= 20 PPC branch, the coefficients and clipping; train_one_epoch must not
    silently replay a step captured for the other phase (engine_proto.py:61-64 switches at epoch 20)."""
    import pytest
    from protopformer_amd.engine import GraphedTrainStep
    early = GraphedTrainStep(model=object(), criterion=None, optimizer=None, epoch=0)
    early.check_matches(epoch=19, ppc_cov_coe=0.1, ppc_mean_coe=0.5, use_ppc_loss=True, max_norm=None)      # same branch: fine
    early.check_matches(epoch=3, ppc_cov_coe=0.3, ppc_mean_coe=0.9, use_ppc_loss=True, max_norm=None)       # coefficients unused before 20
    with pytest.raises(RuntimeError, match="build a new GraphedTrainStep"):
        early.check_matches(epoch=20, ppc_cov_coe=0.1, ppc_mean_coe=0.5, use_ppc_loss=True, max_norm=None)
    late = GraphedTrainStep(model=object(), criterion=None, optimizer=None, epoch=20, ppc_cov_coe=0.1, ppc_mean_coe=0.5)
    late.check_matches(epoch=35, ppc_cov_coe=0.1, ppc_mean_coe=0.5, use_ppc_loss=True, max_norm=None)
    with pytest.raises(RuntimeError):
        late.check_matches(epoch=35, ppc_cov_coe=0.2, ppc_mean_coe=0.5, use_ppc_loss=True, max_norm=None)
    with pytest.raises(RuntimeError):
        late.check_matches(epoch=35, ppc_cov_coe=0.1, ppc_mean_coe=0.5, use_ppc_loss=True, max_norm=1.0)     # captured without clipping
