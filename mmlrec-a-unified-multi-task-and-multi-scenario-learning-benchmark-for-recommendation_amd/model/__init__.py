"""Drop-in model zoo for the hot path: same class names / constructor signatures as the reference's model/*.py."""
from .aitm import AITM  # noqa: F401
from .apg import APG  # noqa: F401
from .basemodel import BaseModel  # noqa: F401
from .cross_stitch import CrossStitch  # noqa: F401
from .escm import ESCM  # noqa: F401
from .esmm import ESMM  # noqa: F401
from .hmoe import HMOE  # noqa: F401
from .mlp import MLP  # noqa: F401
from .mmoe import MMOE  # noqa: F401
from .mssm import MSSM  # noqa: F401
from .pepnet import PepNet  # noqa: F401
from .ple import PLE  # noqa: F401
from .sharedbottom import SharedBottom  # noqa: F401
from .snr_trans import SNR_trans  # noqa: F401
from .star import STAR  # noqa: F401
from .utils import DenseFeat, SparseFeat, VarLenSparseFeat, get_feature_names  # noqa: F401
