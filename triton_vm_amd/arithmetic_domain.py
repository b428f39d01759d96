"""Host mirror of the reference's ArithmeticDomain (/root/reference/triton-vm/src/arithmetic_domain.rs)
whose bulk methods run on the device through the C ABI."""
from . import field
from .capi import Domain


class ArithmeticDomain:
    """offset * <generator>, |<generator>| = length (arithmetic_domain.rs:34-47)."""

    def __init__(self, offset, generator, length):
        self.offset, self.generator, self.length = offset, generator, length

    @classmethod
    def of_length(cls, length):
        """arithmetic_domain.rs:78-85"""
        if length <= 0 or length & (length - 1) or length > 2**32:
            raise ValueError(f"PrimitiveRootNotSupported({length})")
        return cls(field.ONE, field.primitive_root_of_unity(length), length)

    def with_offset(self, offset):
        """arithmetic_domain.rs:89-92"""
        return ArithmeticDomain(offset, self.generator, self.length)

    def __len__(self):
        return self.length

    def pow(self, exponent):
        """arithmetic_domain.rs:280-296"""
        if exponent <= 0 or exponent & (exponent - 1):
            raise ValueError(f"IllegalExponent({exponent})")
        return ArithmeticDomain(field.mont_pow(self.offset, exponent), field.mont_pow(self.generator, exponent),
                                max(self.length // exponent, 1))

    def value(self, n):
        """arithmetic_domain.rs:227-229"""
        return field.mont_mul(field.mont_pow(self.generator, n), self.offset)

    def c(self):
        return Domain(self.offset, self.generator, self.length)

    # -- device methods: arrays are DeviceBuffers of field_kind-word elements ------------------
    def evaluate(self, ctx, d_coeffs, n_coeffs, field_kind=1, out=None):
        """arithmetic_domain.rs:141-170"""
        out = out or ctx.alloc(self.length * field_kind)
        ctx._check(ctx.lib.tvm_evaluate(ctx.handle, field_kind, d_coeffs.ptr if d_coeffs else None, n_coeffs, self.c(),
                                        out.ptr), "tvm_evaluate")
        return out

    def interpolate(self, ctx, d_values, field_kind=1, out=None):
        """arithmetic_domain.rs:182-189"""
        out = out or ctx.alloc(self.length * field_kind)
        ctx._check(ctx.lib.tvm_interpolate(ctx.handle, field_kind, d_values.ptr, self.c(), out.ptr), "tvm_interpolate")
        return out

    def low_degree_extension(self, ctx, d_codeword, target, field_kind=1):
        """arithmetic_domain.rs:203-212"""
        coeffs = self.interpolate(ctx, d_codeword, field_kind)
        return target.evaluate(ctx, coeffs, self.length, field_kind)
