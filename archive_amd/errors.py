"""Error types of the host mirror.

The reference is silent on this path: a corrupt stream yields truncated output, not an
exception.  The only things it can throw here are Dart's RangeError (reads
past the buffer) -- mirrored as RangeError -- and it can fail to terminate on a degenerate
litlen table, which this implementation reports instead of reproducing.
"""


class ArchiveHipError(RuntimeError):
    """libarchive_hip.so reported AHIP_E_* (no device, unsupported construct, bad argument)."""

    def __init__(self, code, message):
        super().__init__("libarchive_hip error %d: %s" % (code, message))
        self.code = code


class RangeError(IndexError):
    """The reference would throw Dart's RangeError on this input (truncated framing,
    back-reference before the start of the output)."""


class ReferenceWouldHang(RuntimeError):
    """The reference's decoder does not terminate on this input (zero-length litlen entry)."""
