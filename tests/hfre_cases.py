"""The seeded HFRE cases live in vlm_fo1_amd/fixtures (bench.py and smoke() use them too); the tests keep importing `hfre_cases`."""
from vlm_fo1_amd.fixtures.hfre_cases import *  # noqa: F401,F403
from vlm_fo1_amd.fixtures.hfre_cases import AUX_DIMS, CASES, DEMO_BOXES, box_fixtures, make_boxes, make_case, pyramid_sizes, smart_grid  # noqa: F401
from vlm_fo1_amd.fixtures.hfre_cases import checksum  # noqa: F401
