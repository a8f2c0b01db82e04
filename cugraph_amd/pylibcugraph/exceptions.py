"""Exception types of the pylibcugraph layer
(reference: python/pylibcugraph/pylibcugraph/exceptions.py:19-26)."""


class FailedToConvergeError(Exception):
    """Raised when an iterative algorithm did not reach its tolerance within max_iterations."""

    pass
