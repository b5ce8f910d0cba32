"""Status codes of include/lsdsort.h as Python exceptions."""
from __future__ import annotations

LSDSORT_OK = 0
LSDSORT_ERR_INVALID_ARG = -1
LSDSORT_ERR_NO_DEVICE = -2
LSDSORT_ERR_HIP = -3
LSDSORT_ERR_WORKSPACE = -4
LSDSORT_ERR_TOO_LARGE = -5
LSDSORT_ERR_UNSUPPORTED = -6
LSDSORT_ERR_DEVICE_FAULT = -7
LSDSORT_ERR_COMM = -8
LSDSORT_ERR_CAPACITY = -9

LSDSORT_ALGO_ONESWEEP = 0
LSDSORT_ALGO_STAGED = 1

LSDSORT_MAX_KEYS = 0x3FFFFFFF


class LsdsortError(RuntimeError):
    """A C-ABI entry returned a negative status.  The reference crashes instead
    (MYCRASH, LSDRadixSort/Utils.h:6-15; CUDA_CALL, LSDRadixSort/CudaUtils.h:7-8)."""

    def __init__(self, status: int, where: str, detail: str = ""):
        self.status = status
        super().__init__(f"{where}: lsdsort status {status}: {detail}")


def check(status: int, where: str) -> None:
    if status != LSDSORT_OK:
        from ._lib import lib

        L = lib()
        detail = L.lsdsort_strerror(status).decode()
        if status == LSDSORT_ERR_HIP:
            detail += f" [{L.lsdsort_last_hip_error()}: {L.lsdsort_last_hip_error_string().decode()}]"
        if status == LSDSORT_ERR_COMM:
            detail += f" [{L.lsdsort_last_comm_error().decode()}]"
        raise LsdsortError(status, where, detail)
