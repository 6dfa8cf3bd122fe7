"""Import shim used ONLY by tests/golden/make_golden.py (never by tests, bench or the product).

It makes the upstream reference at /root/reference importable in the build container by
registering empty stand-ins for optional third-party modules that the reference imports at
module scope but never calls on the ranking path: loguru, faiss, wandb, dgl,
requests.  Nothing here restates reference code; it only unblocks `import rec_pangu`.

/root/reference does not exist on the GPU box: the golden vectors this produces are committed
as plain arrays under tests/golden/*.npz + *.json and are what the tests read.
"""
import sys
import types

REFERENCE_ROOT = "/root/reference"


class _Noop:
    def __getattr__(self, name):
        return _Noop()

    def __call__(self, *a, **k):
        return _Noop()


def install():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("loguru", logger=_Noop())
    mod("faiss")
    mod("wandb", log=_Noop(), init=_Noop(), login=_Noop(), finish=_Noop())
    dgl = mod("dgl", DGLGraph=object)
    dgl.__path__ = []
    fn = mod("dgl.function")
    dgl.function = fn

    def _no_network(*a, **k):
        raise RuntimeError("no network in the build container")

    mod("requests", get=_no_network)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    import rec_pangu  # noqa: F401  (spawns the version-check thread, which fails quietly)
    import rec_pangu.model_pipeline as mp
    import sklearn.metrics as skm
    import numpy as np

    # sklearn>=1.5 dropped log_loss(eps=); the reference (model_pipeline.py:83) still passes it.
    def _log_loss(y_true, y_pred, eps=1e-7, **kw):
        y_pred = np.clip(np.asarray(y_pred, dtype=np.float64), eps, 1 - eps)
        return skm.log_loss(y_true, y_pred, **kw)

    mp.log_loss = _log_loss
    return rec_pangu
