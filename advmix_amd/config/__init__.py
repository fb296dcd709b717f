"""Drop-in for ``lib/config`` (``from config import cfg, update_config``) without yacs,
which is not installable here."""
from .default import _C as cfg, update_config, CfgNode   # noqa: F401
