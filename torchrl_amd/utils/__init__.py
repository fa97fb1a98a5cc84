from .args import get_args, get_params
from .logger import Logger
