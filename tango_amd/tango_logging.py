"""Rank-0-gated, level-thresholded, time-stamped printing (same interface as tango/tango_logging.py:47-131)."""
import datetime
import sys

parallelEnvironment = None
MPIrank = None
threshold = None
initialized = False

CRITICAL, ERROR, WARNING, INFO, DEBUG = 50, 40, 30, 20, 10


def setup(parallel=False, rank=0, level=INFO):
    global parallelEnvironment, MPIrank, threshold, initialized
    if not initialized:
        parallelEnvironment, MPIrank, threshold, initialized = parallel, rank, level, True


def log(message, level=INFO):
    if initialized and serial_or_rank0() and level >= threshold:
        print(add_time_stamp(message))
        sys.stdout.flush()


def debug(message):
    return log(message, DEBUG)


def info(message):
    return log(message, INFO)


def warning(message):
    return log(message, WARNING)


def error(message):
    return log(message, ERROR)


def critical(message):
    return log(message, CRITICAL)


def serial_or_rank0():
    return parallelEnvironment is False or MPIrank == 0


def set_level(level):
    global threshold
    threshold = level


def get_level():
    return threshold


def time_stamp():
    return '{:%H:%M:%S}'.format(datetime.datetime.utcnow())


def add_time_stamp(message):
    return '{} ({})'.format(message, time_stamp())
