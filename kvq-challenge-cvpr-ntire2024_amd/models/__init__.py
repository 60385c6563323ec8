from .model import VQA_Network  # noqa: F401
from .head import VQAHead, simpleVQAHead  # noqa: F401
