"""Import stub for the optional UPN proposal detector.

The reference's `inference.py:3` imports `detect_tools.upn.UPNWrapper` although it never uses it; every
BASELINE configuration takes precomputed proposals (UPN is out of scope for the hot path, ranked
"next" in §8f).  The name resolves so the reference drivers import cleanly; using it raises."""


class UPNWrapper:
    def __init__(self, *args, **kwargs):
        raise NotImplementedError(
            "UPN (Swin-L + deformable DETR proposal detector) is not part of the MI355X hot-path engine; "
            "pass precomputed proposal boxes in message['bbox_list'] (SURVEY.md §8f rank 4).")
