"""COVERM_KNOBS (coverm_amd/csrc/knobs.h): the sizes tests shrink travel in ONE environment variable, "name=value,name=value"."""
import os


def merged(current, **kv):
    """The value of COVERM_KNOBS with `kv` added to (or replacing names in) `current`; a value of None removes the name."""
    have = {}
    for item in (current or "").split(","):
        if "=" in item:
            k, v = item.split("=", 1)
            have[k] = v
    for k, v in kv.items():
        if v is None:
            have.pop(k, None)
        else:
            have[k] = str(v)
    return ",".join("%s=%s" % kv for kv in have.items())


def with_knobs(env, **kv):
    """A copy of the environment dict `env` with the knobs set (for subprocesses)."""
    out = dict(env)
    out["COVERM_KNOBS"] = merged(env.get("COVERM_KNOBS"), **kv)
    return out


def set_knobs(monkeypatch, **kv):
    """Sets the knobs in this process for the duration of a test."""
    monkeypatch.setenv("COVERM_KNOBS", merged(os.environ.get("COVERM_KNOBS"), **kv))
