__version__ = "2.4.0"
from . import functional  # noqa
