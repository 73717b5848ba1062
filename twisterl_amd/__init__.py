"""twisterl_amd -- MI355X-native episode collector for twisteRL (drop-in for the
`twisterl.twisterl.{nn,env,collector}` extension modules on the collection hot path).

    from twisterl_amd import twisterl          # same shape as `from twisterl import twisterl`
    coll = twisterl.collector.PPOCollector(**config["collecting"])
    data = coll.collect(twisterl.env.Puzzle(4, 4, 8, 2, 256), policy)

Everything below the Python surface is hand-written HIP for gfx950 behind a C ABI
(include/twisterl_hip.h); there is no CPU implementation in this package.
"""
from types import SimpleNamespace as _NS

from . import collector, env, nn  # noqa: F401
from ._lib import device_count, device_info, library_path, release_cached_memory  # noqa: F401

# mirror of the PyO3 module tree (python_interface/python_bindings.rs:21-77)
twisterl = _NS(nn=nn, env=env, collector=collector)

__all__ = ["nn", "env", "collector", "twisterl", "device_count", "device_info", "library_path", "release_cached_memory"]
