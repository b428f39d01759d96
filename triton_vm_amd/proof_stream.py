"""Host mirror of the reference's `ProofStream`, `ProofItem`, `Proof` and `Claim`
(/root/reference/triton-vm/src/proof_stream.rs, proof_item.rs:96-150, proof.rs:33-120) with the `BFieldCodec` encoding
and the Fiat-Shamir sampling of `twenty-first = "2.0.0"` [not vendored in the reference tree: restated; pinned by the
reference's proof-digest snapshot, proof.rs:200-226, through tests/test_proof_snapshot.py].

All words are Montgomery words (the `raw_u64` of a BFieldElement), so lengths and discriminants are converted on the way
in and `Proof.words` is the reference's `Vec<BFieldElement>` in memory representation.

BFieldCodec as used here:
  * BFieldElement / XFieldElement / Digest / u32 / [T; N] of those: static length, the elements in order;
  * Vec<T>, T static: [number of elements, elements...]; T dynamic: [number of elements, (len_i, element_i)...];
  * derive on a struct: the fields LAST FIELD FIRST, a dynamically-sized field prefixed with its length;
  * derive on an enum: [discriminant, fields as for a struct];
  * Polynomial: the struct {coefficients: Vec<FF>} without trailing zero coefficients.
"""
import numpy as np

from . import field

# proof_item.rs:96-150, in declaration order (= discriminant): payload kind, in the Fiat-Shamir heuristic?
#   "static": the words as they are; "vec:k": Vec of k-word static elements; "polynomial"; "response": two-field struct
PROOF_ITEMS = [
    ("MerkleRoot", "static", True), ("Log2PaddedHeight", "static", True), ("OutOfDomainMainRow", "static", True),
    ("OutOfDomainAuxRow", "static", True), ("OutOfDomainQuotientSegments", "static", True), ("Polynomial", "polynomial", True),
    ("StirOutOfDomainValues", "vec:3", True), ("AuthenticationStructure", "vec:5", False), ("MasterMainTableRows", "vec:379", False),
    ("MasterAuxTableRows", "vec:273", False), ("QuotientSegmentsElements", "vec:15", False), ("FriCodeword", "vec:3", False),
    ("FriResponse", "response", False), ("StirResponse", "response", False),
]
VARIANT = {name: (k, kind, fs) for k, (name, kind, fs) in enumerate(PROOF_ITEMS)}
# the statically sized payloads (BFieldCodec::static_length): Digest, u32, [XFieldElement; 379], [XFieldElement; 91],
# [XFieldElement; 4] -- a decoder that takes whatever length the proof offers accepts proofs the reference rejects
STATIC_WORDS = {"MerkleRoot": 5, "Log2PaddedHeight": 1, "OutOfDomainMainRow": 379 * 3, "OutOfDomainAuxRow": 91 * 3,
                "OutOfDomainQuotientSegments": 4 * 3}

# the labels the provers in this package enqueue under (longest prefix wins) -> proof item
LABELS = [
    ("log2 padded height", "Log2PaddedHeight"), ("ood main", "OutOfDomainMainRow"), ("ood aux", "OutOfDomainAuxRow"),
    ("ood quot", "OutOfDomainQuotientSegments"), ("fri last codeword", "FriCodeword"), ("fri last polynomial", "Polynomial"),
    ("stir final polynomial", "Polynomial"), ("stir ood values", "StirOutOfDomainValues"), ("fri response", "FriResponse"),
    ("fri auth", "FriResponse"), ("stir response leafs", "StirResponse"), ("stir response auth", "StirResponse"),
    ("main rows", "MasterMainTableRows"), ("aux rows", "MasterAuxTableRows"), ("quot rows", "QuotientSegmentsElements"),
    ("main auth", "AuthenticationStructure"), ("aux auth", "AuthenticationStructure"), ("quot auth", "AuthenticationStructure"),
    ("main root", "MerkleRoot"), ("aux root", "MerkleRoot"), ("quot root", "MerkleRoot"), ("fri root", "MerkleRoot"),
    ("stir root", "MerkleRoot"),
]


def variant_of(label):
    if label in VARIANT:
        return label
    for prefix, variant in LABELS:
        if label.startswith(prefix):
            return variant
    raise KeyError(f"no proof item for the label {label!r}")


# the labels a decoded item (or, for a response, its two parts) is logged under
DECODED_LABELS = {"FriResponse": ("fri response", "fri auth"), "StirResponse": ("stir response leafs", "stir response auth")}


def labels_match(label, expected):
    """does a logged item answer a verifier's request for `expected` (a label prefix or a variant name)?"""
    if label.startswith(expected):
        return True
    variant = variant_of(label)
    if variant != variant_of(expected):
        return False
    return VARIANT[variant][1] != "response" or ("auth" in label) == ("auth" in expected)


class ProofDecodingError(ValueError):
    """ProofStreamError::DecodingError (error.rs)"""


def _m(n):
    return np.uint64(field.to_mont(int(n)))


def _words(a):
    return np.ascontiguousarray(a, dtype=np.uint64).reshape(-1)


def encode_vec(words, elem_words):
    """Vec<T>, T of static length `elem_words`"""
    w = _words(words)
    assert w.size % elem_words == 0
    return np.concatenate([[_m(w.size // elem_words)], w])


def encode_struct(fields):
    """derive(BFieldCodec) on a struct; fields: [(words, is_static)] in declaration order"""
    out = []
    for words, static in reversed(fields):
        if not static:
            out.append(np.array([_m(len(words))], np.uint64))
        out.append(_words(words))
    return np.concatenate(out) if out else np.zeros(0, np.uint64)


def encode_polynomial(coefficients, elem_words=3):
    c = np.ascontiguousarray(coefficients, dtype=np.uint64).reshape(-1, elem_words)
    n = c.shape[0]
    while n and not c[n - 1].any():
        n -= 1
    return encode_struct([(encode_vec(c[:n], elem_words), False)])


def encode_response(leaves, auth_structure):
    """FriResponse {queried_leaves: Vec<XFieldElement>, auth_structure: Vec<Digest>} (fri.rs:101-108) for leaves
    [n][3]; StirResponse {queried_leafs: Vec<Vec<XFieldElement>>, auth_structure} (stir.rs:150-168) for [n][k][3]"""
    leaves = np.ascontiguousarray(leaves, dtype=np.uint64)
    if leaves.ndim == 3:
        parts = [np.array([_m(leaves.shape[0])], np.uint64)]
        for stack in leaves:
            enc = encode_vec(stack, 3)
            parts += [np.array([_m(enc.size)], np.uint64), enc]
        leaves_enc = np.concatenate(parts)
    else:
        leaves_enc = encode_vec(leaves, 3)
    return encode_struct([(leaves_enc, False), (encode_vec(auth_structure, 5), False)])


def encode_item(variant, payload):
    """derive(BFieldCodec) on ProofItem: [discriminant, (payload length if its type is dynamically sized), payload]"""
    k, kind, _ = VARIANT[variant]
    if kind == "static":
        return np.concatenate([[_m(k)], _words(payload)])
    if kind.startswith("vec:"):
        enc = encode_vec(payload, int(kind[4:]))
    elif kind == "polynomial":
        enc = encode_polynomial(payload)
    else:
        enc = encode_response(*payload)
    return np.concatenate([[_m(k), _m(enc.size)], enc])


class Claim:
    """proof.rs:62-120; program_digest [5], input and output: Montgomery words"""

    def __init__(self, program_digest=None, public_input=(), output=(), version=6):
        self.program_digest = _words(np.zeros(5) if program_digest is None else program_digest)
        self.input, self.output, self.version = _words(list(public_input)), _words(list(output)), version

    def encode(self):
        return encode_struct([(self.program_digest, True), (np.array([_m(self.version)], np.uint64), True),
                              (encode_vec(self.input, 1), False), (encode_vec(self.output, 1), False)])


def hash_varlen(lib, words):
    """Tip5::hash_varlen on the host"""
    state = np.zeros(16, np.uint64)
    w = _words(words)
    lib.tvm_host_sponge_pad_and_absorb(state.ctypes.data, w.ctypes.data, w.size)
    return state[:5].copy()


class Proof:
    """proof.rs:38 -- the encoded proof stream"""

    def __init__(self, words):
        self.words = _words(words)

    def encode(self):
        return encode_struct([(encode_vec(self.words, 1), False)])

    def digest(self, lib):
        """Tip5::hash(&proof), canonical values (what the reference's snapshots print)"""
        return [field.from_mont(int(w)) for w in hash_varlen(lib, self.encode())]

    def padded_height(self, lib):
        """Proof::padded_height (proof.rs:45-59): from the one Log2PaddedHeight item"""
        heights = [payload for label, payload, _ in ProofStream.from_proof(lib, self.words).log if variant_of(label) == "Log2PaddedHeight"]
        if not heights:
            raise ProofDecodingError("NoLog2PaddedHeight")
        if len(heights) > 1:
            raise ProofDecodingError("TooManyLog2PaddedHeights")
        return 1 << field.from_mont(int(heights[0][0]))


class ProofStream:
    """proof_stream.rs:8-104.  `enqueue` takes a label (see LABELS) or a ProofItem variant name and the payload words."""

    def __init__(self, lib):
        self.lib = lib
        self.state = np.zeros(16, np.uint64)      # Tip5::init(): the variable-length domain
        self.log = []                             # (label, payload, fiat_shamir) in order: what a verifier dequeues

    def alter_fiat_shamir_state_with(self, words):
        w = _words(words)
        self.lib.tvm_host_sponge_pad_and_absorb(self.state.ctypes.data, w.ctypes.data, w.size)

    def _absorb(self, label, payload):
        self.alter_fiat_shamir_state_with(encode_item(variant_of(label), payload))

    def enqueue(self, label, words, fiat_shamir=None):
        """ProofStream::enqueue (proof_stream.rs:54-59): the item always goes into the proof; it alters the sponge only if
        ProofItem::include_in_fiat_shamir_heuristic says so (proof_item.rs:96-150)"""
        in_heuristic = VARIANT[variant_of(label)][2]
        if fiat_shamir is not None and fiat_shamir != in_heuristic:
            raise ValueError(f"{label!r}: the reference {'includes' if in_heuristic else 'does not include'} this item in the heuristic")
        payload = np.array(words, dtype=np.uint64)
        self.log.append((label, payload, in_heuristic))
        if in_heuristic:
            self._absorb(label, payload)

    @property
    def items(self):
        return [(label, payload.size) for label, payload, _ in self.log]

    def verifier_view(self):
        """a fresh stream over the same items, for a verifier: dequeue() hands out the next item and absorbs it when the
        prover did (ProofStream::dequeue, proof_stream.rs:62-72)"""
        v = ProofStream(self.lib)
        pending = list(self.log)

        def dequeue(expected_prefix=None):
            if not pending:
                raise ValueError("ProofStreamError::EmptyQueue")
            label, payload, in_heuristic = pending.pop(0)
            if expected_prefix is not None and not labels_match(label, expected_prefix):
                raise ValueError(f"unexpected proof item {label!r}, wanted {expected_prefix!r}")
            if in_heuristic:
                v._absorb(label, payload)
            return payload

        v.dequeue = dequeue
        v.pending = pending
        return v

    @classmethod
    def from_proof(cls, lib, proof_words):
        """impl TryFrom<&Proof> for ProofStream (proof_stream.rs:106-113): decode the items; nothing is absorbed yet
        (use verifier_view() to read them the way a verifier does)"""
        w = [field.from_mont(int(x)) for x in _words(proof_words)]
        words = _words(proof_words)
        pos = 0

        def take(n=1):
            nonlocal pos
            if pos + n > len(w):
                raise ProofDecodingError("the proof ends inside an item")
            pos += n
            return pos - n

        def vec(start, length, elem_words):
            """Vec<T> of static elements occupying exactly `length` words from `start` -> payload words"""
            if length < 1 or w[start] * elem_words != length - 1:
                raise ProofDecodingError("a vector's length prefix disagrees with its field length")
            return words[start + 1:start + length]

        self = cls(lib)
        try:
            return self._decode(w, words, take, vec)
        except IndexError:
            raise ProofDecodingError("a length prefix points outside the proof")

    def _decode(self, w, words, take, vec):
        if len(w) < 2 or w[0] != len(w) - 1:
            raise ProofDecodingError("the items field does not span the proof")
        take(1)
        n_items = w[take()]
        for _ in range(n_items):
            size = w[take()]
            start = take(size)
            if size < 1 or w[start] >= len(PROOF_ITEMS):
                raise ProofDecodingError("unknown proof item")
            name, kind, fs = PROOF_ITEMS[w[start]]
            if kind == "static":
                if size - 1 != STATIC_WORDS[name]:
                    raise ProofDecodingError(f"{name}: {size - 1} payload words, the type has {STATIC_WORDS[name]}")
                if name == "Log2PaddedHeight" and w[start + 1] >= 1 << 32:
                    raise ProofDecodingError("Log2PaddedHeight: not a u32")
                self.log.append((name, words[start + 1:start + size].copy(), fs))
                continue
            if size < 2 or w[start + 1] != size - 2:
                raise ProofDecodingError(f"{name}: the payload length disagrees with the item length")
            body, length = start + 2, size - 2
            if kind.startswith("vec:"):
                self.log.append((name, vec(body, length, int(kind[4:])).copy(), fs))
            elif kind == "polynomial":
                if length < 1 or w[body] != length - 1:
                    raise ProofDecodingError("Polynomial: bad field length")
                coefficients = vec(body + 1, length - 1, 3).reshape(-1, 3)
                if len(coefficients) and not coefficients[-1].any():
                    raise ProofDecodingError("Polynomial: trailing zeros in the encoding")
                self.log.append((name, coefficients.copy(), fs))
            else:   # the struct's fields, last field first: auth_structure, then the leaves
                auth_len = w[body]
                auth = vec(body + 1, auth_len, 5)
                leaves_at = body + 1 + auth_len
                leaves_len = w[leaves_at]
                if 2 + auth_len + leaves_len != length:
                    raise ProofDecodingError(f"{name}: field lengths disagree with the item length")
                if name == "FriResponse":
                    leaves = vec(leaves_at + 1, leaves_len, 3).reshape(-1, 3)
                else:   # Vec<Vec<XFieldElement>>
                    stacks, q = [], leaves_at + 2
                    for _ in range(w[leaves_at + 1]):
                        stacks.append(vec(q + 1, w[q], 3).reshape(-1, 3))
                        q += 1 + w[q]
                    if q != leaves_at + 1 + leaves_len:
                        raise ProofDecodingError("StirResponse: the stacks do not span the field")
                    if len({len(st) for st in stacks}) > 1:   # (the verifier rejects ragged stacks; never a numpy error)
                        raise ProofDecodingError("StirResponse: stacks of different heights")
                    height = len(stacks[0]) if stacks else 0   # (explicit shape: no stacks at all is an empty response, not a numpy error)
                    leaves = np.array(stacks, np.uint64).reshape(len(stacks), height, 3)
                self.log.append((DECODED_LABELS[name][0], leaves.copy(), fs))
                self.log.append((DECODED_LABELS[name][1], auth.copy(), fs))
        if take(0) != len(w):
            raise ProofDecodingError("words after the last item")
        return self

    # -- Fiat-Shamir sampling [twenty-first Tip5::sample_scalars / sample_indices] -----------------------------
    def _squeeze(self):
        out = self.state[:10].copy()
        self.lib.tvm_host_tip5_permutation(self.state.ctypes.data)
        return out

    def sample_scalars(self, n):
        words = np.concatenate([self._squeeze() for _ in range((3 * n + 9) // 10)])
        return words[:3 * n].reshape(n, 3)

    def sample_indices(self, upper_bound, n):
        out, pending = [], []
        while len(out) < n:
            if not pending:
                pending = list(self._squeeze())
            v = field.from_mont(int(pending.pop(0)))
            if v != field.P - 1:
                out.append(v % upper_bound)
        return out

    # -- the proof ------------------------------------------------------------------------------------------------
    def encoded_items(self):
        """the ProofItems in order; a response's two parts (leaves, authentication structure) are one item"""
        out, k = [], 0
        while k < len(self.log):
            label, payload, _ = self.log[k]
            variant = variant_of(label)
            if VARIANT[variant][1] == "response":
                auth_label, auth, _ = self.log[k + 1]
                assert variant_of(auth_label) == variant and "auth" in auth_label, (label, auth_label)
                out.append(encode_item(variant, (payload, auth)))
                k += 2
            else:
                out.append(encode_item(variant, payload))
                k += 1
        return out

    def proof(self):
        """impl From<&ProofStream> for Proof (proof_stream.rs:115-119): the struct's only encoded field is `items`"""
        items = self.encoded_items()
        parts = [np.array([_m(len(items))], np.uint64)]
        for enc in items:
            parts += [np.array([_m(enc.size)], np.uint64), enc]
        return Proof(encode_struct([(np.concatenate(parts), False)]))
