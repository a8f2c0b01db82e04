"""reference: python/cugraph/cugraph/exceptions.py:19-24"""


class FailedToConvergeError(Exception):
    """Raised when an algorithm fails to converge within a predetermined set of constraints
    which vary based on the algorithm, and may or may not be user-configurable."""
