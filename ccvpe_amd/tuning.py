"""Persistence of the library's tuning table (include/ccvpe.h: ccvpe_import_tuning / ccvpe_export_tuning).

The first forward of a new (batch, ground size) measures every tiled launch with each candidate kernel tile; which tile wins
can differ between two processes, and with it the last bits of the result.  A plan found in the table is not measured: it
runs the recorded launches, so every process that loads the same table computes the same bits and starts in milliseconds.

Two sources, loaded in this order (the later one wins per plan):
  * `ccvpe_amd/tuning/gfx950.txt` - the table committed with the library: the plans of the BASELINE workloads as measured on
    an MI355X (regenerate with `python tools/make_tuning_table.py` on a GPU box);
  * the user cache `$CCVPE_TUNE_CACHE` (a file; default `~/.cache/ccvpe_amd/tuning-<library digest>.txt`; `0` / `off` / empty
    disables it): plans this machine has measured itself; written back (merged, atomically, under a lock) whenever a handle
    has tuned a new plan.  Best effort: an unwritable location costs the re-tune on the next start, nothing else.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

HERE = os.path.dirname(os.path.abspath(__file__))
COMMITTED = os.path.join(HERE, "tuning", "gfx950.txt")


def user_cache_path() -> Optional[str]:
    env = os.environ.get("CCVPE_TUNE_CACHE")
    if env is not None:
        return None if env.strip().lower() in ("", "0", "off", "none") else env
    from . import _lib
    return os.path.join(os.path.expanduser("~"), ".cache", "ccvpe_amd", f"tuning-{_lib.library_digest()[:12]}.txt")


def parse(text: str) -> Dict[str, str]:
    """launch key -> 'op <key> <tile> <split>' line; anything else in the text is dropped."""
    table: Dict[str, str] = {}
    for line in text.splitlines():
        parts = line.split()
        if len(parts) == 4 and parts[0] == "op" and parts[3].isdigit():
            table[parts[1]] = line.strip()
    return table


def render(table: Dict[str, str]) -> str:
    return "".join(table[k] + "\n" for k in sorted(table))


def load_into(lib, handle) -> int:
    """Import the committed table and the user cache into a fresh handle; returns the number of launches read."""
    n = 0
    committed = None if os.environ.get("CCVPE_TUNE_IGNORE_COMMITTED") else COMMITTED   # tests: start from an empty table
    for path in (committed, user_cache_path()):
        if not path or not os.path.exists(path):
            continue
        try:
            with open(path) as fh:
                text = render(parse(fh.read()))   # normalised: foreign lines never reach the library
        except OSError:
            continue
        rc = lib.ccvpe_import_tuning(handle, text.encode())
        if rc > 0:
            n += rc
    return n


def export(lib, handle) -> str:
    need = C.c_size_t(0)
    lib.ccvpe_export_tuning(handle, None, 0, C.byref(need))
    buf = C.create_string_buffer(max(need.value, 1))
    if lib.ccvpe_export_tuning(handle, buf, len(buf), None) != 0:
        return ""
    return buf.value.decode()


def save_from(lib, handle, path: Optional[str] = None) -> Optional[str]:
    """Merge the handle's table into the user cache (entries of the handle replace same-key entries of the file)."""
    path = path or user_cache_path()
    if not path:
        return None
    mine = parse(export(lib, handle))
    if not mine:
        return None
    try:
        import fcntl
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path + ".lock", "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                merged: Dict[str, str] = {}
                if os.path.exists(path):
                    with open(path) as fh:
                        merged = parse(fh.read())
                merged.update(mine)
                tmp = f"{path}.tmp.{os.getpid()}"
                with open(tmp, "w") as fh:
                    fh.write(render(merged))
                os.replace(tmp, path)
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
        return path
    except OSError:
        return None
