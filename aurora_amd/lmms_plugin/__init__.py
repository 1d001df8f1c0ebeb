"""lmms-eval plugin package for the MI355X AuroraCap path.

    LMMS_EVAL_PLUGINS=aurora_amd.lmms_plugin  python -m lmms_eval --model auroracap_mi355x \
        --model_args pretrained=<xtuner-format dir>,token_merge_ratio=0.4,batch_size=16 --tasks vdc ...

lmms-eval (src/lmms-eval/lmms_eval/models/__init__.py:62-70) imports `<plugin>.models`, reads its AVAILABLE_MODELS
and imports `<plugin>.models.<name>`; the class registers itself with @register_model (api/registry.py:11-24).
"""
