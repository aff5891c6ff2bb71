"""Built-in scenarios, loadable by file name like the reference's ``scenarios.load``
(multiagent/scenarios/__init__.py:5-7): ``load("simple_spread.py").Scenario()``."""
import importlib
import importlib.util
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def load(name):
    stem = name[:-3] if name.endswith(".py") else name
    path = os.path.join(_HERE, stem + ".py")
    if os.path.exists(path) and stem != "__init__":
        return importlib.import_module(__name__ + "." + stem)
    if os.path.exists(name):                      # a user's scenario file given by path
        spec = importlib.util.spec_from_file_location("mpe_user_scenario_" + stem.replace(os.sep, "_"), name)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    raise FileNotFoundError("no scenario %r (built in: collector, consensus, simple, simple_adversary, simple_crypto, simple_predator_prey, simple_push, simple_reference, simple_speaker_listener, simple_spread, simple_tag, simple_world_comm)" % (name,))
