"""rmt_app_amd - MI355X-native drop-in for PyREMOT's dynamic packed-bed reactor path (model N2).

    from rmt_app_amd import rmtExe, rmtCom      # same names as `from PyREMOT import rmtExe, rmtCom`
"""
from .rmt import rmtCom, rmtExe  # noqa: F401
from .settings import MODEL_SETTING, solverSetting  # noqa: F401

__version__ = "0.1.0"
