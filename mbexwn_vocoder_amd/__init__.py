"""MI355X-native mel-inversion path of the Multi-Band Excited WaveNet vocoder.

Drop-in for the inference surface of roebel/MBExWN_Vocoder (``MBExWN_NVoc``):
``MELInverter`` / ``list_models`` / ``get_config_file`` / ``mbexwn_version`` keep the names, argument
meaning and error behaviour of reference MBExWN_NVoc/__init__.py:12-65 and MBExWN_NVoc/mel_inverter.py.
"""
import copy
import os
from typing import Union

mbexwn_version = (1, 2, 3)   # version of the reference surface this package mirrors (reference __init__.py:12)

# reference __init__.py:19-31 -- domain -> ordered list of model names (first entry = default of the domain)
_mel_inv_models = {
    "SING": [
        "MBExWN_SIIConv_V71g_SING_IMP0_IMPORTmod_MCFG0_WNCHA320_DCHA32_1024_DPTACT0_ADLW0.1_GMCFG5_24kHz",
    ],
    "SPEECH": [
        "MBExWN_SIIConv_V71g_SPEECH_IMP0_IMPORTmod_MCFG0_WNCHA320_DCHA32_1024_DPTACT0_ADLW0.1_GMCFG5_24kHz",
    ],
    "VOICE": [
        "MBExWN_SIIConv_V71g_VOICE2_WNCHA340_IMP0_WNCHA340_IMPORTmod_MCFG0_WNCHA340_DCHA32_1024_DPTACT0_ADLW0.1_GMCFG0_24kHz",
    ],
}


def list_models(voice_type: Union[str, None] = None):
    """Dictionary of the available mel-inverter models per voice domain (reference __init__.py:33-44;
    like the reference, ``voice_type`` does not filter)."""
    return copy.deepcopy(_mel_inv_models)


def models_root():
    """Directory that holds one sub-directory per model (config.yaml + weights).  The reference keeps it
    inside the package (``MBExWN_NVoc/models``); ``MBEXWN_MODELS_DIR`` overrides it."""
    return os.environ.get("MBEXWN_MODELS_DIR", os.path.join(os.path.dirname(os.path.abspath(__file__)), "models"))


def get_config_file(model_id_or_path, verbose=False):
    """Resolve a model id (any sub-string of ``<DOMAIN>/<model name>``) or a model directory to its
    ``config.yaml`` (reference __init__.py:47-65).

    Deviation from the reference, which it documents as defects: the first matching
    model wins (the reference's inner-only ``break`` makes the last one win) and an unknown id raises
    ``FileNotFoundError`` instead of ``UnboundLocalError``.
    """
    model_dir = None
    if os.path.exists(model_id_or_path):
        model_dir = model_id_or_path
    else:
        for domain, names in list_models().items():
            for name in names:
                if model_id_or_path in f"{domain}/{name}":
                    model_dir = os.path.join(models_root(), name)
                    break
            if model_dir is not None:
                break
    if model_dir is None:
        raise FileNotFoundError(f"error::no model matches id {model_id_or_path}")
    config_file = os.path.join(model_dir, "config.yaml")
    if not os.path.exists(config_file):
        raise FileNotFoundError(f"error::loading config file from {config_file}")
    return config_file


from .mel_inverter import MELInverter  # noqa: E402,F401
