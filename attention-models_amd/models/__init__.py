"""Drop-in alias: put ``attention-models_amd/`` on PYTHONPATH in place of the reference
checkout and ``from models import SoftmaxAttention, ViTVQGAN, ...`` resolves to the
MI355X-native classes (same names as /root/reference/models/__init__.py)."""
from amk.models import *  # noqa: F401,F403
from amk.models import __all__  # noqa: F401
