import typing


class Writer:
    writer: 'Writer'

    def __init__(
        self,
        index_file_path: str,
        max_chunk_len: typing.Optional[int] = None,
        *,
        device: typing.Optional[int] = None,
        devices: typing.Optional[typing.Sequence[int]] = None,
        format_version: int = 1,
        striped: bool = False,
    ) -> None: ...

    def add_entries_from_file_lines(self, input_file_path: str) -> None: ...

    def add_entry(self, text: str) -> None: ...

    def dump_data(self) -> None: ...

    def finalize(self) -> None: ...

    def close(self) -> None: ...

    def __enter__(self) -> 'Writer': ...

    def __exit__(self, *exc: typing.Any) -> None: ...


class PackedResult(typing.NamedTuple):
    data: typing.Any      # numpy uint8 array: all entries back to back
    offsets: typing.Any   # numpy uint64 array [num_entries + 1]
    counts: typing.Any    # numpy uint64 array [num_queries]


class DeviceResult(typing.NamedTuple):
    data: typing.Any      # torch uint8 tensor in HBM: all entries back to back
    starts: typing.Any    # torch int64 tensor [num_entries]: start of every entry
    counts: typing.Any    # torch int64 tensor [num_queries]
    num_bytes: int


class Reader:
    reader: 'Reader'

    def __init__(
        self,
        index_file_path: str,
        *,
        device: typing.Optional[int] = None,
        devices: typing.Optional[typing.Sequence[int]] = None,
        shard: typing.Tuple[int, int] = (0, 1),
        order: typing.Optional[str] = None,
    ) -> None: ...

    def set_result_order(self, order: str) -> None: ...

    @property
    def result_order(self) -> str: ...

    @property
    def num_chunks(self) -> int: ...

    @property
    def residency(self) -> typing.Dict[str, int]: ...

    def search(self, substring: str) -> typing.List[str]: ...

    def search_multiple(self, substrings: typing.List[str]) -> typing.List[str]: ...

    def count(self, substring: str) -> int: ...

    def count_multiple(self, substrings: typing.List[str]) -> typing.List[int]: ...

    def search_batch_raw(
        self, patterns: typing.Sequence[bytes],
    ) -> typing.Tuple[typing.List[bytes], typing.List[int]]: ...

    def search_batch_packed(self, patterns: typing.Sequence[bytes]) -> PackedResult: ...

    def search_batch_device(self, patterns: typing.Sequence[bytes]) -> DeviceResult: ...

    def search_multiple_bytes_as_str(self, patterns: typing.Sequence[bytes]) -> typing.List[str]: ...

    def count_multiple_bytes(self, patterns: typing.Sequence[bytes]) -> typing.List[int]: ...

    def last_stats(self) -> typing.Dict[str, float]: ...

    def close(self) -> None: ...

    def __enter__(self) -> 'Reader': ...

    def __exit__(self, *exc: typing.Any) -> None: ...


def device_count() -> int: ...


def release_workspace() -> None: ...
def workspace_bytes(device: int = ...) -> int: ...
