"""yacs-compatible subset: attribute+item access, freeze/defrost, YAML merge with
``literal_eval`` of strings (``GPUS: (0,1,...)`` -> tuple), trailing ``KEY VAL`` overrides.
Default tree = lib/config/default.py:17-139; ``update_config`` = :143-184."""
import ast
import copy
import os

import yaml


class CfgNode(dict):
    def __init__(self, init=None, new_allowed=False):
        super().__init__()
        object.__setattr__(self, '_frozen', False)
        object.__setattr__(self, '_new_allowed', new_allowed)
        for k, v in (init or {}).items():
            self[k] = CfgNode(v, new_allowed=True) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        if object.__getattribute__(self, '_frozen'):
            raise AttributeError('Attempted to set {} to {}, but CfgNode is immutable'.format(k, v))
        self[k] = v

    def _walk(self, fn):
        fn(self)
        for v in self.values():
            if isinstance(v, CfgNode):
                v._walk(fn)

    def freeze(self):
        self._walk(lambda n: object.__setattr__(n, '_frozen', True))

    def defrost(self):
        self._walk(lambda n: object.__setattr__(n, '_frozen', False))

    def clone(self):
        return copy.deepcopy(self)

    @staticmethod
    def _decode(v):
        if isinstance(v, str):
            try:
                return ast.literal_eval(v)
            except (ValueError, SyntaxError):
                return v
        return v

    def _merge(self, other, path=''):
        for k, v in other.items():
            if isinstance(v, dict):
                if k not in self:
                    if not object.__getattribute__(self, '_new_allowed'):
                        raise KeyError('Non-existent config key: {}{}'.format(path, k))
                    self[k] = CfgNode(new_allowed=True)
                if isinstance(self[k], CfgNode):
                    self[k]._merge(v, path + k + '.')
                else:
                    self[k] = CfgNode(v, new_allowed=True)
            else:
                if k not in self and not object.__getattribute__(self, '_new_allowed'):
                    raise KeyError('Non-existent config key: {}{}'.format(path, k))
                self[k] = self._decode(v)

    def merge_from_file(self, path):
        with open(path) as f:
            self._merge(yaml.safe_load(f) or {})

    def merge_from_list(self, opts):
        opts = list(opts or [])
        assert len(opts) % 2 == 0, 'Override list has odd length: {}'.format(opts)
        for key, val in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split('.')
            for p in parts[:-1]:
                node = node[p]
            if parts[-1] not in node and not object.__getattribute__(node, '_new_allowed'):
                raise KeyError('Non-existent config key: {}'.format(key))
            node[parts[-1]] = self._decode(val)

    def __deepcopy__(self, memo):
        n = CfgNode(new_allowed=object.__getattribute__(self, '_new_allowed'))
        for k, v in self.items():
            n[k] = copy.deepcopy(v, memo)
        return n


_DEFAULTS = {
    'OUTPUT_DIR': '', 'LOG_DIR': '', 'DATA_DIR': '', 'GPUS': (0,), 'WORKERS': 4, 'PRINT_FREQ': 20,
    'AUTO_RESUME': False, 'PIN_MEMORY': True, 'RANK': 0,
    'CUDNN': {'BENCHMARK': True, 'DETERMINISTIC': False, 'ENABLED': True},
    'MODEL': {'NAME': 'pose_hrnet', 'INIT_WEIGHTS': True, 'PRETRAINED': '', 'NUM_JOINTS': 17,
              'TAG_PER_JOINT': True, 'TARGET_TYPE': 'gaussian', 'IMAGE_SIZE': [256, 256],
              'HEATMAP_SIZE': [64, 64], 'SIGMA': 2, 'EXTRA': {}},
    'LOSS': {'USE_OHKM': False, 'TOPK': 8, 'USE_TARGET_WEIGHT': True, 'USE_DIFFERENT_JOINTS_WEIGHT': False},
    'DATASET': {'ROOT': '', 'ROOT_C': '', 'DATASET': 'mpii', 'TRAIN_SET': 'train', 'TEST_SET': 'valid',
                'DATA_FORMAT': 'jpg', 'HYBRID_JOINTS_TYPE': '', 'SELECT_DATA': False, 'FLIP': True,
                'SCALE_FACTOR': 0.25, 'ROT_FACTOR': 30, 'PROB_HALF_BODY': 0.0, 'NUM_JOINTS_HALF_BODY': 8,
                'COLOR_RGB': False, 'MINI_COCO': False, 'VAL_FG': False, 'VAL_MASK': False,
                'VAL_PARSING': False},
    'TRAIN': {'LR_FACTOR': 0.1, 'LR_STEP': [90, 110], 'LR': 0.001, 'OPTIMIZER': 'adam', 'MOMENTUM': 0.9,
              'WD': 0.0001, 'NESTEROV': False, 'GAMMA1': 0.99, 'GAMMA2': 0.0, 'BEGIN_EPOCH': 0,
              'END_EPOCH': 140, 'RESUME': False, 'CHECKPOINT': '', 'BATCH_SIZE_PER_GPU': 32, 'SHUFFLE': True},
    'TEST': {'BATCH_SIZE_PER_GPU': 32, 'FLIP_TEST': False, 'POST_PROCESS': False, 'SHIFT_HEATMAP': False,
             'USE_GT_BBOX': False, 'TEST_ROBUST': False, 'CORRUPTION_TYPE': '', 'IMAGE_THRE': 0.1,
             'NMS_THRE': 0.6, 'SOFT_NMS': False, 'OKS_THRE': 0.5, 'IN_VIS_THRE': 0.0, 'COCO_BBOX_FILE': '',
             'BBOX_THRE': 1.0, 'MODEL_FILE': '', 'MASK_FILE': '', 'SOFT_ARGMAX': False, 'BIAS': 0.0},
    'DEBUG': {'DEBUG': False, 'SAVE_BATCH_IMAGES_GT': False, 'SAVE_BATCH_IMAGES_PRED': False,
              'SAVE_HEATMAPS_GT': False, 'SAVE_HEATMAPS_PRED': False},
}


def _build(d, new_allowed=False):
    n = CfgNode(new_allowed=new_allowed)
    for k, v in d.items():
        n[k] = _build(v, new_allowed=(k == 'EXTRA')) if isinstance(v, dict) else v
    return n


_C = _build(_DEFAULTS)
_C['TEST']['SEVERITY'] = 0


def update_config(cfg, args):
    cfg.defrost()
    cfg.merge_from_file(args.cfg)
    cfg.merge_from_list(getattr(args, 'opts', None))
    if getattr(args, 'modelDir', None):
        cfg.OUTPUT_DIR = args.modelDir
    if getattr(args, 'logDir', None):
        cfg.LOG_DIR = args.logDir
    if getattr(args, 'dataDir', None):
        cfg.DATA_DIR = args.dataDir
    if getattr(args, 'corruption_type', None):
        cfg.TEST.CORRUPTION_TYPE = args.corruption_type
    cfg.TEST.SEVERITY = getattr(args, 'severity', 0)
    cfg.TEST.TEST_ROBUST = getattr(args, 'test_robust', False)
    cfg.DATASET.ROOT = os.path.join(cfg.DATA_DIR, cfg.DATASET.ROOT)
    cfg.DATASET.ROOT_C = os.path.join(cfg.DATA_DIR, 'data/coco-C' if cfg.DATASET.DATASET == 'coco' else 'data/mpii-C')
    cfg.MODEL.PRETRAINED = os.path.join(cfg.DATA_DIR, cfg.MODEL.PRETRAINED)
    if cfg.TEST.MODEL_FILE:
        cfg.TEST.MODEL_FILE = os.path.join(cfg.DATA_DIR, cfg.TEST.MODEL_FILE)
    cfg.freeze()
