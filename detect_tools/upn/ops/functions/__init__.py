from .ms_deform_attn_func import MSDeformAttnFunction  # noqa: F401  (reference: ops/functions/__init__.py)
