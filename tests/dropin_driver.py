"""Build-container only (needs /root/reference; never travels to the GPU box): runs the reference's OWN tools/train.py
main() - parse_args, update_config, get_pose_net, UnetGenerator, the three DataParallel wraps, JointsMSELoss,
get_optimizer, --load_from_D, MultiStepLR - with the sys.modules swap of INTEGRATION.md section 2 applied, up to the
first call into the training loop (the first device work), and prints what arrived there as one JSON line.

usage: dropin_driver.py <yaml> <tmpdir> <checkpoint.pth|-> [--plain]

Stubbed for this container (none of it is on the hot path): torchvision.transforms, tensorboardX, the reference's
``dataset`` package (cv2 / json_tricks / pycocotools are absent) and ``.cuda()`` (no GPU here).
"""
import importlib
import json
import os
import runpy
import sys
import types

REF = '/root/reference'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Reached(Exception):
    pass


def main():
    yaml_path, tmp, ckpt = sys.argv[1:4]
    plain = '--plain' in sys.argv[4:]
    import torch
    sys.path.insert(0, ROOT)

    # ---- container stubs (not part of the recipe) --------------------------------------------------------------
    tv = types.ModuleType('torchvision')
    tvt = types.ModuleType('torchvision.transforms')
    tvt.Compose = lambda fs: ('compose', fs)
    tvt.ToTensor = lambda: 'to_tensor'
    tvt.Normalize = lambda mean, std: ('normalize', mean, std)
    tv.transforms = tvt
    sys.modules['torchvision'], sys.modules['torchvision.transforms'] = tv, tvt
    tbx = types.ModuleType('tensorboardX')

    class SummaryWriter:
        def __init__(self, log_dir=None):
            self.log_dir = log_dir

        def add_graph(self, *a, **k):
            raise RuntimeError('no graph in the stub')       # tools/train.py:97-100 swallows it

        def add_scalar(self, *a, **k):
            pass

        def close(self):
            pass
    tbx.SummaryWriter = SummaryWriter
    sys.modules['tensorboardX'] = tbx
    ds = types.ModuleType('dataset')

    class _DS(torch.utils.data.Dataset):
        def __init__(self, cfg, args, root, image_set, is_train, transform=None):
            self.is_train = is_train

        def __len__(self):
            return 64

        def __getitem__(self, i):
            raise RuntimeError('the drop-in test stops before any batch is loaded')
    ds.coco = ds.mpii = _DS
    sys.modules['dataset'] = ds
    torch.nn.Module.cuda = lambda self, device=None: self
    torch.Tensor.cuda = lambda self, *a, **k: self

    # ---- INTEGRATION.md section 2, verbatim: what a maintainer appends to tools/_init_paths.py -----------------
    sys.path.insert(0, os.path.join(REF, 'tools'))
    import _init_paths                                      # noqa: F401  (the reference's: puts lib/ on sys.path)
    import advmix_amd                                       # noqa: F401
    for pkg in ('models', 'core', 'nms', 'config'):
        sys.modules[pkg] = importlib.import_module('advmix_amd.' + pkg)
        for sub in ('pose_hrnet', 'pose_resnet', 'Unet_generator', 'function', 'loss', 'evaluate',
                    'inference', 'nms', 'default'):
            try:
                sys.modules['%s.%s' % (pkg, sub)] = importlib.import_module('advmix_amd.%s.%s' % (pkg, sub))
            except ModuleNotFoundError:
                pass
    import advmix_amd.utils.utils as _u
    import utils.utils as ref_utils                         # the reference's logger / summary helpers stay
    assert ref_utils.__file__.startswith(REF), ref_utils.__file__
    ref_utils.get_optimizer = _u.get_optimizer
    ref_utils.save_checkpoint = _u.save_checkpoint
    ref_utils.create_logger = _u.create_logger              # N-process safe (the reference's mkdir races between ranks)
    from advmix_amd.dp import Replica
    torch.nn.DataParallel = Replica                         # one process per GPU: DataParallel's shape, none of its mechanics
    torch.nn.parallel.DataParallel = Replica
    from advmix_amd.dp import ShardedDataLoader, rank0_only
    torch.utils.data.DataLoader = ShardedDataLoader         # per-rank shard + per-GPU batch under a process group; DataLoader otherwise
    torch.save = rank0_only(torch.save)                     # tools/train.py:337 writes final_state.pth unguarded
    import shutil
    from advmix_amd.dp import atomic_copy
    shutil.copy2 = atomic_copy(shutil.copy2)                # tools/train.py:73-84: N ranks copy three files to ONE directory
    import tensorboardX
    tensorboardX.SummaryWriter = _u.rank0_summary_writer(tensorboardX.SummaryWriter)    # :87-91: one event file, rank 0's

    # ---- "the only edit multi-GPU needs" (INTEGRATION.md section 2): the process group, before main() builds anything.
    # gloo stands in for nccl here (no GPU in this container).
    if int(os.environ.get('WORLD_SIZE', '1')) > 1:
        import torch.distributed as dist
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%s' % os.environ['MASTER_PORT'],
                                rank=int(os.environ['RANK']), world_size=int(os.environ['WORLD_SIZE']))

    # ---- the probe: the loops are where the first device call would happen -------------------------------------
    import advmix_amd.core.function as F_
    got = {}

    def probe_advmix(config, args, train_loader, models, criterion, optimizers, epoch, output_dir, tb_log_dir,
                     writer_dict, grad_sync=None):
        got.update(kind='train_advmix', config=config, args=args, loader=train_loader, models=models,
                   criterion=criterion, optimizers=optimizers, epoch=epoch, output_dir=output_dir, tb_log_dir=tb_log_dir,
                   writer=writer_dict['writer'])
        raise Reached()

    def probe_plain(config, args, train_loader, model, criterion, optimizer, epoch, output_dir, tb_log_dir,
                    writer_dict, grad_sync=None):
        got.update(kind='train', config=config, args=args, loader=train_loader, models=[model], criterion=criterion,
                   optimizers=[optimizer], epoch=epoch, output_dir=output_dir)
        raise Reached()
    real = (F_.train_advmix, F_.train)
    F_.train_advmix, F_.train = probe_advmix, probe_plain

    argv = ['tools/train.py', '--cfg', yaml_path]
    if not plain:
        argv += ['--advmix']
    if ckpt != '-':
        argv += ['--load_from_D', ckpt]
    if 'w48' in os.path.basename(yaml_path):
        argv += ['--downsamples', '5']                      # the 384x288 generator (tools/_init_parse.py:132-134)
    argv += ['OUTPUT_DIR', os.path.join(tmp, 'output'), 'LOG_DIR', os.path.join(tmp, 'log'), 'WORKERS', '0',
             'DATA_DIR', tmp]               # MODEL.PRETRAINED resolves under DATA_DIR (config/default.py update_config)
    if int(os.environ.get('WORLD_SIZE', '1')) > 1:          # one rank per listed GPU, as the YAMLs' GPUS: (0,...,7) means
        argv += ['GPUS', '(%s,)' % ','.join(str(i) for i in range(int(os.environ['WORLD_SIZE'])))]
    sys.argv = argv
    os.chdir(REF)                                           # train.py copies 'tools/train.py' relative to the cwd (:81-83)
    try:
        runpy.run_path(os.path.join(REF, 'tools', 'train.py'), run_name='__main__')
        raise SystemExit('tools/train.py returned without reaching the training loop')
    except Reached:
        pass
    if plain:                                               # tools/train.py:288 prints optimizer_G's lr unconditionally:
        raise SystemExit('unreachable')                     # the reference itself cannot run without --advmix

    # ---- what arrived at the loop ------------------------------------------------------------------------------
    import advmix_amd.models as pm
    cfg = got['config']
    models = got['models']
    D = models[0]
    assert isinstance(D, Replica) and all(isinstance(m, Replica) for m in models)
    inner = D.module
    assert type(inner).__module__.startswith('advmix_amd.models.'), type(inner)
    out = {'kind': got['kind'], 'model_class': type(inner).__module__ + '.' + type(inner).__name__,
           'keys': sorted(D.state_dict().keys()), 'epoch': got['epoch'],
           'criterion': type(got['criterion']).__module__, 'use_target_weight': got['criterion'].use_target_weight,
           'optimizers': [type(o).__module__ + '.' + type(o).__name__ for o in got['optimizers']],
           'lrs': [o.param_groups[0]['lr'] for o in got['optimizers']],
           'n_params': [sum(len(g['params']) for g in o.param_groups) for o in got['optimizers']],
           'batch_size': got['loader'].batch_size, 'gpus': list(cfg.GPUS),
           'sampler': type(got['loader'].sampler).__name__, 'loader_len': len(got['loader']),
           'shard': sorted(int(i) for i in got['loader'].sampler)[:4] + [len(list(got['loader'].sampler))],
           'real_loop_signature_ok': True, 'output_dir_files': sorted(os.listdir(got['output_dir'])),
           'output_dir': got['output_dir'], 'tb_log_dir': got.get('tb_log_dir'),
           'writer': type(got.get('writer')).__name__, 'logger_level': __import__('logging').getLogger().level}
    if len(models) == 3:
        G, T = models[1], models[2]
        assert type(G.module) is pm.Unet_generator.UnetGenerator
        out['g_keys'] = len(G.state_dict())
        out['g_downs'] = got['args'].downsamples
        out['teacher_is_copy'] = T.module is not inner and all(
            bool((a == b).all()) for a, b in zip(T.state_dict().values(), D.state_dict().values()))
    pre = cfg.MODEL.PRETRAINED
    if pre and os.path.isfile(pre):                         # the ImageNet checkpoint get_pose_net -> init_weights loaded
        psd, msd = torch.load(pre), D.state_dict()
        out['pretrained_keys'] = len(psd)
        out['pretrained_loaded_into_D'] = sorted(k for k, v in psd.items() if 'module.' + k in msd
                                                 and msd['module.' + k].shape == v.shape
                                                 and bool((msd['module.' + k] == v).all()))
        fl = msd['module.final_layer.weight']
        out['final_layer_std'] = float(fl.std())
        out['final_layer_bias_absmax'] = float(msd['module.final_layer.bias'].abs().max())
    if ckpt != '-':
        sd = torch.load(ckpt)
        msd, tsd = D.state_dict(), (models[2].state_dict() if len(models) == 3 else None)
        loaded = [k for k, v in sd.items() if 'module.' + k in msd and msd['module.' + k].shape == v.shape
                  and bool((msd['module.' + k] == v).all())]
        out['ckpt_keys'] = len(sd)
        out['ckpt_loaded_into_D'] = len(loaded)
        if tsd is not None:
            out['ckpt_loaded_into_teacher'] = sum(
                1 for k, v in sd.items() if 'module.' + k in tsd and tsd['module.' + k].shape == v.shape
                and bool((tsd['module.' + k] == v).all()))
    # the real loop accepts exactly these arguments (signature check against the un-probed functions)
    import inspect
    for fn, n in zip(real, (10, 10)):
        ps = list(inspect.signature(fn).parameters)
        out['real_loop_signature_ok'] &= ps[:n] == ['config', 'args', 'train_loader', 'models' if fn is real[0] else 'model',
                                                     'criterion', 'optimizers' if fn is real[0] else 'optimizer', 'epoch',
                                                     'output_dir', 'tb_log_dir', 'writer_dict']
    print('DROPIN ' + json.dumps(out))


if __name__ == '__main__':
    main()
