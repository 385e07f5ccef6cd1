"""Model plugin discovery: `create_model(opt)` resolves opt['model'] to `models/<name>_model.py` ->
class `<Name>Model` by the reference's filename + lower-case class-name convention
(codes/models/__init__.py:8-75; aliases srgan|srragan -> sr at :59-60)."""
import importlib
import logging
import os

logger = logging.getLogger("base")


def find_model(model_name):
    folder = os.path.dirname(os.path.abspath(__file__))
    files = [os.path.splitext(f)[0] for f in sorted(os.listdir(folder)) if f.endswith("_model.py")]
    wanted = "{}_model".format(model_name).lower()
    match = [f for f in files if f.lower() == wanted]
    if not match:
        raise NotImplementedError("Model [{:s}] not recognized by the HIP engine (no {}.py).".format(model_name, wanted))
    module = importlib.import_module("{}.{}".format(__name__, match[0]))
    target = "{}model".format(model_name.replace("_", "")).lower()
    for name, cls in vars(module).items():
        if name.lower() == target and isinstance(cls, type):
            return cls
    raise NotImplementedError("Model [{:s}] not recognized: {} has no class named {} (case-insensitive).".format(
        model_name, match[0], target))


def create_model(opt, step=0, verbose=True):
    model = opt["model"]
    if model in ("srgan", "srragan"):
        model = "sr"
    instance = find_model(model)(opt)
    if verbose:
        logger.info("Model [%s] created.", instance.__class__.__name__)
    return instance
