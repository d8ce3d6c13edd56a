"""`make -C dir -s` under an exclusive file lock, so that test processes running side by side (pytest -n) do not
rebuild the same helper library at the same time.  Test-only."""
import fcntl
import os
import subprocess


def locked_make(d: str) -> None:
    with open(os.path.join(d, ".make.lock"), "w") as f:
        fcntl.flock(f, fcntl.LOCK_EX)
        try:
            subprocess.check_call(["make", "-C", d, "-s"])
        finally:
            fcntl.flock(f, fcntl.LOCK_UN)
