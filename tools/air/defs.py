"""Column / challenge namespaces with master-table indices
(/root/reference/triton-air/src/table.rs:27-103, table_column.rs:506-748, challenge_id.rs)."""
from types import SimpleNamespace

from . import names

MAIN, AUX = {}, {}
_m = _a = 0
MAIN_START, AUX_START = {}, {}
for _t in names.TABLES:
    MAIN_START[_t], AUX_START[_t] = _m, _a
    MAIN[_t] = SimpleNamespace(**{n: _m + i for i, n in enumerate(names.MAIN_COLUMNS[_t])})
    AUX[_t] = SimpleNamespace(**{n: _a + i for i, n in enumerate(names.AUX_COLUMNS[_t])})
    _m += len(names.MAIN_COLUMNS[_t])
    _a += len(names.AUX_COLUMNS[_t])
NUM_MAIN_COLUMNS, NUM_AUX_COLUMNS = _m, _a          # 149, 49 (table.rs:29-52)
Ch = SimpleNamespace(**{n: i for i, n in enumerate(names.CHALLENGES)})
NUM_CHALLENGES = len(names.CHALLENGES)              # 63
TARGET_DEGREE = 4                                    # triton-air/src/lib.rs:37

TIP5_RATE = 10
TIP5_STATE_SIZE = 16
TIP5_NUM_ROUNDS = 5
TIP5_NUM_SPLIT_AND_LOOKUP = 4
DIGEST_LEN = 5
EXTENSION_DEGREE = 3

# cross_table_argument.rs:40-90 default initials
PERM_ARG_INITIAL = 1
EVAL_ARG_INITIAL = 1
LOOKUP_ARG_INITIAL = 0
