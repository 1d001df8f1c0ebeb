"""Model table read by lmms-eval's plugin loader (lmms_eval/models/__init__.py:62-70): module name -> class name."""
AVAILABLE_MODELS = {"auroracap_mi355x": "AuroraCapMI355X"}
