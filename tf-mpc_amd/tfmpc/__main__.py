"""``python -m tfmpc ...`` = the command line of ``tf-mpc_amd/scripts/tfmpc.py`` (reference: the
``tfmpc`` console script of ``setup.py:44-47``)."""
import importlib.util
import os

_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scripts", "tfmpc.py")
_spec = importlib.util.spec_from_file_location("tfmpc_cli", _path)
_cli = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_cli)

if __name__ == "__main__":
    _cli.cli()
