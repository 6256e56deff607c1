"""CPU oracle for the swimmer / ARS hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product package never does (tests/test_no_oracle_in_product.py enforces it).
"""
from .swimmer_oracle import (  # noqa: F401
    OracleParams, build, accelerations, step, reset, rollout, step_batch, rollout_batch,
    num_threads, set_num_threads, cpu_share, twin_accelerations, twin_step, twin_step_batch, twin_system,
)
