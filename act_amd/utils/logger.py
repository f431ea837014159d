"""rank-0 logging helpers (reference: utils/logger.py:32-130), without mmcv/termcolor."""
import logging

import torch.distributed as dist

logger_initialized = {}


def get_logger(name, log_file=None, log_level=logging.INFO, file_mode="w"):
    logger = logging.getLogger(name)
    if name in logger_initialized:
        return logger
    for logger_name in logger_initialized:
        if name.startswith(logger_name):
            return logger
    logger.propagate = False
    handlers = [logging.StreamHandler()]
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    if rank == 0 and log_file is not None:
        handlers.append(logging.FileHandler(log_file, file_mode))
    fmt = logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s")
    for h in handlers:
        h.setFormatter(fmt)
        h.setLevel(log_level)
        logger.addHandler(h)
    logger.setLevel(log_level if rank == 0 else logging.ERROR)
    logger_initialized[name] = True
    return logger


def get_root_logger(log_file=None, log_level=logging.INFO, name="main"):
    return get_logger(name=name, log_file=log_file, log_level=log_level)


def print_log(msg, logger=None, level=logging.INFO):
    if logger is None:
        print(msg)
    elif isinstance(logger, logging.Logger):
        logger.log(level, msg)
    elif logger == "silent":
        pass
    elif isinstance(logger, str):
        get_logger(logger).log(level, msg)
    else:
        raise TypeError(f'logger should be either a logging.Logger object, str, "silent" or None, but got {type(logger)}')
