"""detect_tools.upn.ops — the UPN detector's native operator on MI355X (reference: detect_tools/upn/ops/, a CUDA extension built by
its own setup.py; here a kernel of libfo1hip.so)."""
