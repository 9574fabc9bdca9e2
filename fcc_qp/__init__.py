# Drop-in package name of the reference (/root/reference/fcc_qp/__init__.py:1 is
# `from fcc_qp_solver import *`): `from fcc_qp import FCCQP, FCCQPSolution, FCCQPOptions`
# (reference fcc_qp_test.py:6) keeps working, now backed by the MI355X HIP solver.
from fcc_qp_amd.fcc_qp_solver import *  # noqa: F401,F403
from fcc_qp_amd.fcc_qp_solver import FCCQP, FCCQPBatch, FCCQPDetails, FCCQPOptions, FCCQPSolution  # noqa: F401
