"""TEST INFRASTRUCTURE (not part of the product): `ProofStream::try_from(&Proof)` and the verifier's side of the transcript,
restated with the oracle's own pieces only -- the BFieldCodec layout of oracle/real_prover.py's encoder read backwards, and its
Tip5 sponge (oracle/tvm_oracle.c's permutation).  The restated Verifier::verify (oracle/real_verifier.py, oracle/ldt_verifier.py)
reads proofs through this view, so that accepting a device proof involves no line of the product's proof_stream.py and no call
into the product's host-side Tip5.

Reference: /root/reference/triton-vm/src/proof_stream.rs:8-113 (dequeue absorbs an item iff the prover's enqueue did),
proof_item.rs:96-150 (variants, their payload types, membership in the Fiat-Shamir heuristic), proof.rs:38.
"""
import numpy as np

from . import oracle as orc
from .real_prover import IN_FIAT_SHAMIR, VARIANTS_OF_PROOF_ITEM, Sponge

P = (1 << 64) - (1 << 32) + 1
STATIC_WORDS = {"MerkleRoot": 5, "Log2PaddedHeight": 1, "OutOfDomainMainRow": 379 * 3, "OutOfDomainAuxRow": 91 * 3,
                "OutOfDomainQuotientSegments": 4 * 3}
VEC_ELEMENT_WORDS = {"StirOutOfDomainValues": 3, "AuthenticationStructure": 5, "MasterMainTableRows": 379,
                     "MasterAuxTableRows": 273, "QuotientSegmentsElements": 15, "FriCodeword": 3}
# the names the restated verifiers ask under -> (variant, part of a response item)
REQUESTS = {"fri root": ("MerkleRoot", None), "stir root": ("MerkleRoot", None), "fri last codeword": ("FriCodeword", None),
            "fri last polynomial": ("Polynomial", None), "stir final polynomial": ("Polynomial", None),
            "stir ood values": ("StirOutOfDomainValues", None), "fri response": ("FriResponse", 0), "fri auth": ("FriResponse", 1),
            "stir response leafs": ("StirResponse", 0), "stir response auth": ("StirResponse", 1)}


class DecodingError(ValueError):
    """ProofStreamError::DecodingError"""


def _vec(values, words, start, length, elem_words):
    """Vec<T> of static elements in exactly `length` words from `start` -> the payload (Montgomery words)"""
    if length < 1 or values[start] * elem_words != length - 1:
        raise DecodingError("a vector's length prefix disagrees with its field length")
    return words[start + 1:start + length]


def decode(proof_words):
    """the Montgomery words of a Proof -> [(variant, canonical encoding of the item, payload parts)]"""
    words = np.ascontiguousarray(proof_words, dtype=np.uint64).reshape(-1)
    v = [int(x) for x in orc.from_mont(words)] if words.size else []
    if len(v) < 2 or v[0] != len(v) - 1:
        raise DecodingError("the items field does not span the proof")
    items, pos = [], 2
    for _ in range(v[1]):
        if pos >= len(v):
            raise DecodingError("the proof ends inside the item list")
        size = v[pos]
        start, end = pos + 1, pos + 1 + size
        if size < 1 or end > len(v) or v[start] >= len(VARIANTS_OF_PROOF_ITEM):
            raise DecodingError("unknown or truncated proof item")
        name = VARIANTS_OF_PROOF_ITEM[v[start]]
        encoding = v[start:end]
        if name in STATIC_WORDS:
            if size - 1 != STATIC_WORDS[name] or (name == "Log2PaddedHeight" and v[start + 1] >= 1 << 32):
                raise DecodingError(f"{name}: payload of the wrong static length")
            parts = (words[start + 1:end],)
        else:
            if size < 2 or v[start + 1] != size - 2:
                raise DecodingError(f"{name}: the payload length disagrees with the item length")
            body, length = start + 2, size - 2
            if name in VEC_ELEMENT_WORDS:
                parts = (_vec(v, words, body, length, VEC_ELEMENT_WORDS[name]),)
            elif name == "Polynomial":
                if length < 1 or v[body] != length - 1:
                    raise DecodingError("Polynomial: bad field length")
                coefficients = _vec(v, words, body + 1, length - 1, 3).reshape(-1, 3)
                if len(coefficients) and not coefficients[-1].any():
                    raise DecodingError("Polynomial: trailing zeros in the encoding")
                parts = (coefficients,)
            else:   # {Fri,Stir}Response: the struct's fields last field first -- auth_structure, then the leaves
                auth_len = v[body]
                auth = _vec(v, words, body + 1, auth_len, 5)
                leaves_at = body + 1 + auth_len
                if leaves_at >= end:
                    raise DecodingError(f"{name}: no leaves field")
                leaves_len = v[leaves_at]
                if 2 + auth_len + leaves_len != length:
                    raise DecodingError(f"{name}: field lengths disagree with the item length")
                if name == "FriResponse":
                    leaves = _vec(v, words, leaves_at + 1, leaves_len, 3).reshape(-1, 3)
                else:
                    stacks, q = [], leaves_at + 2
                    for _k in range(v[leaves_at + 1]):
                        stacks.append(_vec(v, words, q + 1, v[q], 3).reshape(-1, 3))
                        q += 1 + v[q]
                    if q != leaves_at + 1 + leaves_len or len({len(st) for st in stacks}) > 1:
                        raise DecodingError("StirResponse: the stacks do not span the field or differ in height")
                    leaves = np.array(stacks, np.uint64).reshape(len(stacks), -1, 3)
                parts = (leaves, auth)
        items.append((name, encoding, parts))
        pos = end
    if pos != len(v):
        raise DecodingError("words after the last item")
    return items


class VerifierView:
    """what Verifier::verify holds: the decoded items in order and the Fiat-Shamir sponge"""

    def __init__(self, proof_words):
        self.sponge = Sponge()
        self.pending = []
        for name, encoding, parts in decode(proof_words):
            for k, part in enumerate(parts):   # a response is asked for in two steps; it is absorbed (never: not in the heuristic) once
                self.pending.append((name, encoding if k == 0 else None, k if len(parts) > 1 else None, part))

    def alter_fiat_shamir_state_with(self, words_mont):
        self.sponge.pad_and_absorb_all([int(x) for x in orc.from_mont(np.ascontiguousarray(words_mont, dtype=np.uint64).reshape(-1))])

    def dequeue(self, request):
        if not self.pending:
            raise ValueError("ProofStreamError::EmptyQueue")
        want, part = REQUESTS.get(request, (request, None))
        name, encoding, k, payload = self.pending.pop(0)
        if name != want or (part is not None and k != part):
            raise ValueError(f"unexpected proof item {name!r}, wanted {request!r}")
        if encoding is not None and name in IN_FIAT_SHAMIR:
            self.sponge.pad_and_absorb_all(encoding)
        return payload

    def sample_scalars(self, n):
        return np.array(self.sponge.sample_scalars(n), np.uint64).reshape(n, 3)

    def sample_indices(self, upper_bound, n):
        return self.sponge.sample_indices(upper_bound, n)
