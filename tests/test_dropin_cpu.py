"""The drop-in claim, executed: the reference's own tools/train.py wiring runs on this repo's
mirrors after the sys.modules swap of INTEGRATION.md section 2 - for the three YAMLs the reference ships and the two
this repo adds - up to the first call into the training loop.  Build-container only: /root/reference does not exist
on the GPU box (and nothing from it is copied: the driver executes the reference's file where it lies)."""
import json
import os
import subprocess
import sys

import pytest
import torch

from helpers import gold_json

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, 'tools', 'train.py')),
                                reason='needs the reference tree (build container only)')

YAMLS = {
    'ref_coco_res50': (REF + '/experiments/coco/resnet/res50_256x192_d256x3_adam_lr1e-3_advmix.yaml', 'resnet50', 17, 6),
    'ref_mpii_res50': (REF + '/experiments/mpii/resnet/res50_256x256_d256x3_adam_lr1e-3_advmix.yaml', 'resnet50', 16, 6),
    'ref_mpii_w32': (REF + '/experiments/mpii/hrnet/w32_256x256_adam_lr1e-3_advmix.yaml', 'hrnet_w32', 16, 6),
    'own_coco_w32': (ROOT + '/experiments/coco/hrnet/w32_256x192_adam_lr1e-3_advmix.yaml', 'hrnet_w32', 17, 6),
    'own_coco_w48': (ROOT + '/experiments/coco/hrnet/w48_384x288_adam_lr1e-3_advmix.yaml', 'hrnet_w48', 17, 5),
}


def _reference_format_checkpoint(path, arch):
    """A ``final_state.pth`` as the reference writes it (tools/train.py:337: model.module.state_dict(), no ``module.``
    prefix): the REAL reference's keys and shapes (tests/golden/state_dict_keys.json, J = 17), random values."""
    g = torch.Generator().manual_seed(5)
    sd = {}
    for k, shape in gold_json('state_dict_keys.json')[arch]:
        sd[k] = torch.zeros((), dtype=torch.int64) if k.endswith('num_batches_tracked') else \
            torch.rand(shape, generator=g) + 0.5
    torch.save(sd, path)
    return sd


def _imagenet_checkpoint(tmp_path, yaml_path, arch):
    """The ImageNet checkpoint MODEL.PRETRAINED names (resolved under DATA_DIR), in the upstream format: the trunk's
    keys under the pose network's own names plus the classifier's keys the pose network does not have."""
    import yaml
    with open(yaml_path) as f:
        y = yaml.safe_load(f)
    path = tmp_path / y['MODEL']['PRETRAINED']
    path.parent.mkdir(parents=True, exist_ok=True)
    g = torch.Generator().manual_seed(9)
    sd = {}
    for k, shape in gold_json('state_dict_keys.json')[arch]:
        if k.split('.')[0] in ('final_layer', 'deconv_layers'):
            continue
        sd[k] = torch.zeros((), dtype=torch.int64) if k.endswith('num_batches_tracked') else \
            torch.rand(shape, generator=g) + 2.0
    if 'hrnet' in arch:
        sd['incre_modules.0.0.conv1.weight'] = torch.rand(32, 32, 1, 1)
        sd['final_layer.0.weight'] = torch.rand(2048, 1024, 1, 1)       # the classifier's, another shape: filtered out
        sd['classifier.weight'] = torch.rand(1000, 2048)
    else:
        sd['fc.weight'], sd['fc.bias'] = torch.rand(1000, 2048), torch.rand(1000)
    torch.save(sd, str(path))
    return sd, y['MODEL']['EXTRA'].get('PRETRAINED_LAYERS', ['*'])


def _drive(yaml_path, tmp_path, ckpt):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'dropin_driver.py'), yaml_path, str(tmp_path), ckpt],
                         capture_output=True, text=True, timeout=600, env=dict(os.environ, PYTHONPATH=ROOT))
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('DROPIN ')]
    assert out.returncode == 0 and lines, (out.stdout[-1500:], out.stderr[-3000:])
    return json.loads(lines[-1][7:])


@pytest.mark.parametrize('name', ['ref_coco_res50', 'own_coco_w32'])
def test_init_weights_loads_the_imagenet_checkpoint_like_the_reference(name, tmp_path):
    """get_pose_net(cfg, is_train=True) -> init_weights(cfg.MODEL.PRETRAINED) as tools/train.py:60 reaches it (no
    --load_from_D on top): the trunk comes from the checkpoint (HRNet: filtered by PRETRAINED_LAYERS, pose_hrnet.py:480-
    489; ResNet: strict=False, pose_resnet.py:236-237), heads keep N(0, 1e-3) / zero bias, foreign keys are ignored."""
    yaml_path, arch, J, downs = YAMLS[name]
    psd, layers = _imagenet_checkpoint(tmp_path, yaml_path, arch)
    r = _drive(yaml_path, tmp_path, '-')
    own = {k for k, _ in gold_json('state_dict_keys.json')[arch]}
    want = sorted(k for k in psd if k in own and (layers[0] == '*' or k.split('.')[0] in layers)
                  and not k.startswith('final_layer'))
    assert r['pretrained_keys'] == len(psd) and r['pretrained_loaded_into_D'] == want and len(want) > 100
    assert 5e-4 < r['final_layer_std'] < 2e-3 and r['final_layer_bias_absmax'] == 0.0
    assert r['teacher_is_copy']                              # copy.deepcopy(model), tools/train.py:65


@pytest.mark.parametrize('name', sorted(YAMLS))
def test_reference_train_py_runs_on_the_mirrors_up_to_the_loop(name, tmp_path):
    yaml_path, arch, J, downs = YAMLS[name]
    ckpt = str(tmp_path / 'final_state.pth')
    sd = _reference_format_checkpoint(ckpt, arch)
    _imagenet_checkpoint(tmp_path, yaml_path, arch)
    r = _drive(yaml_path, tmp_path, ckpt)
    assert r['kind'] == 'train_advmix' and r['epoch'] == 0
    assert r['model_class'].startswith('advmix_amd.models.pose_' + ('hrnet' if 'hrnet' in arch else 'resnet'))
    # state-dict keys = the reference's, with the prefix DataParallel adds (what :198-235 / AUTO_RESUME rely on)
    assert r['keys'] == sorted('module.' + k for k, _ in gold_json('state_dict_keys.json')[arch])    # (names; J only changes final_layer's shape)
    assert r['g_keys'] == len(gold_json('state_dict_keys.json')['unet%d' % downs]) and r['g_downs'] == downs
    # --load_from_D went through the reference's own ``module.`` + size filter: everything but a J-dependent head
    skipped = 0 if J == 17 else 2
    assert r['ckpt_keys'] == len(sd)
    assert r['ckpt_loaded_into_D'] == len(sd) - skipped and r['ckpt_loaded_into_teacher'] == len(sd) - skipped
    assert r['criterion'] == 'advmix_amd.core.loss' and r['use_target_weight'] is True
    assert r['optimizers'] == ['advmix_amd.utils.utils.FlatAdam'] * 2 and r['lrs'] == [1e-3, 1e-3]
    assert r['n_params'][0] == sum(1 for k, s in gold_json('state_dict_keys.json')[arch] if 'running_' not in k
                                   and not k.endswith('num_batches_tracked'))
    assert r['batch_size'] == 32 * len(r['gpus']) and r['real_loop_signature_ok']
    # create_logger, the three shutil.copy2 calls (:72-83): the reference's own side effects happened
    assert 'train.py' in r['output_dir_files'] and os.path.basename(yaml_path) in r['output_dir_files']


def _run_ranks(world, yaml_path, tmp_path, port):
    procs = []
    for r in range(world):
        env = dict(os.environ, PYTHONPATH=ROOT, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   OMP_NUM_THREADS='1', MKL_NUM_THREADS='1')
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'dropin_driver.py'), yaml_path, str(tmp_path), '-'],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    res = []
    for p in procs:
        so, se = p.communicate(timeout=900)
        lines = [ln for ln in so.splitlines() if ln.startswith('DROPIN ')]
        assert p.returncode == 0 and lines, (so[-1500:], se[-3000:])
        res.append(json.loads(lines[-1][7:]))
    return res


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _check_shared_run(res, world, tmp_path, yaml_path):
    """What N ranks of the unmodified main() must agree on, and the single-writer side effects."""
    for i, r in enumerate(res):
        assert r['kind'] == 'train_advmix' and r['gpus'] == list(range(world))
        assert r['batch_size'] == 32 and r['sampler'] == 'DistributedSampler'
        assert r['shard'][-1] == 64 // world and r['optimizers'] == ['advmix_amd.utils.utils.FlatAdam'] * 2
    # every rank got rank 0's directories (one time stamp, published through the store), and only rank 0 a real writer
    assert len({r['output_dir'] for r in res}) == 1 and len({r['tb_log_dir'] for r in res}) == 1
    assert [r['writer'] for r in res] == ['SummaryWriter'] + ['NullSummaryWriter'] * (world - 1)
    assert len({tuple(r['shard'][:4]) for r in res}) == world
    out = res[0]['output_dir']
    files = sorted(os.listdir(out))
    assert 'train.py' in files and os.path.basename(yaml_path) in files and 'pose_hrnet.py' in files
    assert not [f for f in files if '.tmp.' in f]                       # dp.atomic_copy left nothing behind
    with open(os.path.join(out, 'train.py'), 'rb') as a, open(os.path.join(REF, 'tools', 'train.py'), 'rb') as b:
        assert a.read() == b.read()                                     # whole, whichever rank's rename came last
    logs = [f for f in files if f.endswith('.log')]
    assert len(logs) == world and sum('_rank' in f for f in logs) == world - 1     # one writer per log file


def test_reference_train_py_on_two_ranks_gets_sharded_loaders(tmp_path):
    """with one process per GPU the reference's unmodified main() must not load the GLOBAL batch on every rank.
    Two gloo ranks run tools/train.py (process group created first - the one edit INTEGRATION.md names - and
    dp.ShardedDataLoader bound as torch.utils.data.DataLoader by the recipe): each arrives at train_advmix with the
    per-GPU batch (32), a DistributedSampler over its own half of the data set, Replica-wrapped models and the flat
    optimizers.  VERDICT r5 weak 6: the reference's own create_logger raced here (both ranks pass ``exists()``, the second
    ``mkdir()`` raises - 1 run in 5); the recipe now binds utils.utils.create_logger / shutil.copy2 / SummaryWriter to
    N-process-safe twins, and this test holds what they promise (tools/loop_dropin_ranks.sh: 50 / 50 at two ranks)."""
    yaml_path, arch, J, downs = YAMLS['own_coco_w32']
    _imagenet_checkpoint(tmp_path, yaml_path, arch)
    res = _run_ranks(2, yaml_path, tmp_path, _free_port())
    _check_shared_run(res, 2, tmp_path, yaml_path)
    assert all(r['loader_len'] == 1 for r in res)                       # 64 samples / 2 ranks / 32


def test_reference_train_py_on_eight_ranks_fresh_output_dir(tmp_path):
    """The first real 8-GPU run: eight ranks, a FRESH OUTPUT_DIR / LOG_DIR (nothing exists: every rank would try to make
    the directories).  GPUS = (0..7), global batch 256 -> 32 per rank over 64 stub samples = 8 per rank."""
    yaml_path, arch, J, downs = YAMLS['own_coco_w32']
    _imagenet_checkpoint(tmp_path, yaml_path, arch)
    res = _run_ranks(8, yaml_path, tmp_path, _free_port())
    _check_shared_run(res, 8, tmp_path, yaml_path)
