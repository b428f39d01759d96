"""Host-side scalar helpers over F_p (python ints).  Plumbing for building domains and arguments;
no table or codeword ever goes through here."""
P = 2**64 - 2**32 + 1
R = 2**64 % P
R_INV = pow(R, -1, P)
ONE = R  # Montgomery word of 1


def to_mont(v):
    return v % P * R % P


def from_mont(raw):
    return raw * R_INV % P


def mont_mul(a, b):
    return a * b % P * R_INV % P


def mont_pow(a, e):
    return to_mont(pow(from_mont(a), e, P))


def mont_inv(a):
    return to_mont(pow(from_mont(a), -1, P))


def generator():
    """[twenty-first, not in the reference tree] BFieldElement::generator() = 7."""
    return to_mont(7)


def primitive_root_of_unity(order):
    """[twenty-first, not in the reference tree] the 2^32-th root 7^((p-1)/2^32) squared down to `order`."""
    assert order and order & (order - 1) == 0 and order <= 2**32
    return to_mont(pow(7, (P - 1) // order, P))
