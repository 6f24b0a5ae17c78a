"""MI355X-native drop-in for the trust-region subproblem path of
antonior92/ip-nonlinear-solver (``ipsolver``).

Public names mirror the reference (ipsolver/__init__.py:3-6).
"""
