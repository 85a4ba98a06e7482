"""Scenarios for the operator-registry / platform surface, run twice with the SAME code: by
tests/golden/gen_fixtures.py (`operator_registry` group) against the reference's `sparsevllm.operators.registry`,
`sparsevllm.operators.decode_attention` and `sparsevllm.platforms`, and by tests/test_operator_registry.py against this
build's modules of the same names.  Every scenario returns plain JSON data: chosen provider names, `rejected` tuples,
exception classes and texts, dataclass field tables, launch configurations.  The cases follow the reference's own
tests/test_operator_registry.py and tests/test_platforms.py (priority order, rejection diagnostics, duplicate names,
equal-priority tie-break, constructor kwargs, empty registry, runtime stats aggregation, CPU platform, unknown platform).
Contains no reference code: `ns` is the namespace of the implementation under test.
"""

from __future__ import annotations

import dataclasses
import os


def _err(fn):
    try:
        return {"ok": fn()}
    except Exception as e:      # class + text are the contract
        return {"err": type(e).__name__, "msg": str(e)}


def _fields(cls):
    out = []
    for f in dataclasses.fields(cls):
        default = None if f.default is dataclasses.MISSING else repr(f.default)
        out.append([f.name, default])
    return out


def _caps(ns, **kw):
    base = dict(platform=ns.PlatformEnum.CUDA, device_type="cuda", device_index=0, device_name="test", compute_capability=(9, 0))
    base.update(kw)
    return ns.DeviceCaps(**base)


@dataclasses.dataclass(frozen=True)
class Spec:
    enabled: bool = True


def _provider(ns, name, priority, verdict, *, init=None):
    """A provider class in the reference's pattern; `verdict(spec, caps)` -> SupportResult."""
    body = {"name": name, "priority": priority, "supports": classmethod(lambda cls, spec, caps: verdict(spec, caps))}
    if init is not None:
        body["__init__"] = init
    return type(name.title(), (), body)


def scenario_registry(ns):
    R, S = ns.OpRegistry, ns.SupportResult
    out = {}

    # highest supporting priority wins; the chosen provider is bound under the registry's family
    reg = R("_test")
    reg.register(_provider(ns, "portable", 10, lambda s, c: S.yes()))
    reg.register(_provider(ns, "specialized", 20, lambda s, c: S.yes() if s.enabled else S.no("disabled")))
    saved = dict(ns.bindings)
    ns.bindings.clear()
    try:
        res = ns.OpResolver(reg).resolve(Spec(), _caps(ns))
        out["priority"] = {"chosen": res.provider.name, "rejected": [list(x) for x in res.rejected],
                           "bound": res.provider in ns.bindings["_test"], "families": sorted(ns.bindings),
                           "result_fields": _fields(type(res))}
        res2 = ns.OpResolver(reg).resolve(Spec(enabled=False), _caps(ns))
        out["priority_disabled"] = {"chosen": res2.provider.name, "rejected": [list(x) for x in res2.rejected]}
    finally:
        ns.bindings.clear()
        ns.bindings.update(saved)
    out["family"] = reg.family
    out["providers"] = [p.name for p in reg.providers]

    # every rejection is reported
    reg = R("_test")
    reg.register(_provider(ns, "first", 10, lambda s, c: S.no("wrong dtype")))
    reg.register(_provider(ns, "second", 20, lambda s, c: S.no("wrong architecture")))
    out["all_rejected"] = _err(lambda: ns.OpResolver(reg).resolve(Spec(), _caps(ns)).provider.name)

    # duplicate names
    reg = R("_dup")
    reg.register(_provider(ns, "same", 1, lambda s, c: S.yes()))
    out["duplicate"] = _err(lambda: reg.register(_provider(ns, "same", 2, lambda s, c: S.yes())).name)

    # equal priority -> name order
    reg = R("_test")
    reg.register(_provider(ns, "zulu", 10, lambda s, c: S.yes()))
    reg.register(_provider(ns, "alpha", 10, lambda s, c: S.yes()))
    out["equal_priority"] = ns.OpResolver(reg).resolve(Spec(), _caps(ns)).provider.name

    # fallback keeps the diagnostics
    reg = R("_test")
    reg.register(_provider(ns, "specialized", 100, lambda s, c: S.no("missing optional library")))
    reg.register(_provider(ns, "portable", 10, lambda s, c: S.yes()))
    res = ns.OpResolver(reg).resolve(Spec(), _caps(ns))
    out["fallback"] = {"chosen": res.provider.name, "rejected": [list(x) for x in res.rejected]}

    # constructor kwargs
    def init(self, marker):
        self.marker = marker
    reg = R("_test")
    reg.register(_provider(ns, "configured", 1, lambda s, c: S.yes(), init=init))
    out["ctor_kwargs"] = ns.OpResolver(reg).resolve(Spec(), _caps(ns), marker="ready").provider.marker
    out["ctor_kwargs_missing"] = _err(lambda: ns.OpResolver(reg).resolve(Spec(), _caps(ns)).provider.marker)

    # empty registry
    out["empty"] = _err(lambda: ns.OpResolver(R("_empty")).resolve(Spec(), _caps(ns)).provider.name)

    # SupportResult
    out["support_result"] = {"fields": _fields(S), "yes": dataclasses.asdict(S.yes()), "yes_reason": dataclasses.asdict(S.yes("fine")),
                             "no": dataclasses.asdict(S.no("because")),
                             "frozen": "err" in _err(lambda: setattr(S.yes(), "supported", False))}
    return out


def scenario_runtime_stats(ns):
    class Provider:
        name = "composite"

        def __init__(self, eager, captured, fallback):
            self.eager, self.captured, self.fallback = eager, captured, fallback

        def runtime_kernel_stats(self):
            return {"kernel_paths": {"tilelang_score": {"eager_dispatches": self.eager,
                                                        "cuda_graph_capture_dispatches": self.captured}},
                    "fallback_reasons": {"noncontiguous:output": self.fallback}}

    class Plain:
        def __init__(self, implementation_name):
            self.implementation_name = implementation_name

    saved = dict(ns.bindings)
    ns.bindings.clear()
    try:
        keep = [Provider(2, 1, 0), Provider(3, 4, 1), Plain("triton"), Plain("flashinfer_sm90"), Plain("triton")]
        ns.record_operator_binding("MLA attention", keep[0])
        ns.record_operator_binding("MLA attention", keep[1])
        ns.record_operator_binding("Attention", keep[2])
        ns.record_operator_binding("block-scaled FP8 Linear", keep[3])
        ns.record_operator_binding("block-scaled FP8 Linear", keep[4])
        stats = ns.operator_runtime_stats()
        del keep[1]
        import gc
        gc.collect()
        after = ns.operator_runtime_stats()           # bindings are weak: a dead provider drops out
        return {"stats": stats, "after_one_died": after["MLA attention"]}
    finally:
        ns.bindings.clear()
        ns.bindings.update(saved)


def scenario_version(ns):
    f = ns.runtime_version_at_least
    table = [(None, (12, 0)), ("12.4", (12, 0)), ("12.4", (12, 4)), ("12.4", (12, 5)), ("11.8.1", (12, 0)), (" 7.2.0", (7, 0)),
             ("6.4.43482-0f2d60242", (6, 4)), ("abc", (1, 0)), ("13", (12, 0)), ("13.0", (12, 9))]
    return [[v, list(m), bool(f(v, m))] for v, m in table]


def scenario_types(ns):
    caps = _caps(ns)
    return {"device_caps_fields": _fields(ns.DeviceCaps),
            "device_caps_frozen": "err" in _err(lambda: setattr(caps, "device_name", "x")),
            "platform_enum": [m.name for m in ns.PlatformEnum],
            "allocator_stats_fields": _fields(ns.AllocatorStats),
            "platform_public_api": sorted(n for n in dir(ns.Platform) if not n.startswith("_")),
            "platform_defaults": {"name": ns.Platform.name, "device_type": ns.Platform.device_type,
                                  "enum": ns.Platform.enum.name,
                                  "distributed_backend": ns.Platform().get_distributed_backend(),
                                  "attention_backend": ns.Platform().get_default_attention_backend(),
                                  "check_available": ns.Platform().check_available(),
                                  "validate_environment": _err(lambda: ns.Platform().validate_environment()),
                                  "caps": [[f.name, repr(getattr(ns.Platform().get_device_caps(2), f.name))]
                                           for f in dataclasses.fields(ns.DeviceCaps)][:12]}}


def scenario_decode_launch(ns):
    import torch
    Spec_ = ns.DecodeAttentionLaunchSpec
    out = {"spec_fields": _fields(Spec_)}
    bad = [dict(num_query_heads=0, num_kv_heads=1, head_dim=128), dict(num_query_heads=4, num_kv_heads=-1, head_dim=128),
           dict(num_query_heads=6, num_kv_heads=4, head_dim=128), dict(num_query_heads=4, num_kv_heads=2, head_dim=0),
           dict(num_query_heads=4, num_kv_heads=2, head_dim=64, page_size=0)]
    out["spec_errors"] = [_err(lambda kw=kw: repr(Spec_(activation_dtype=torch.bfloat16, **kw))) for kw in bad]
    out["spec_frozen"] = "err" in _err(lambda: setattr(Spec_(4, 2, 64, torch.bfloat16), "head_dim", 1))
    out["registry_family"] = ns.DECODE_ATTENTION_LAUNCH_REGISTRY.family
    out["has_default_provider"] = "default_gqa" in [p.name for p in ns.DECODE_ATTENTION_LAUNCH_REGISTRY.providers]
    D = ns.DefaultGqaDecodeLaunchProvider
    out["default"] = {"name": D.name, "priority": D.priority,
                      "supports": dataclasses.asdict(D.supports(Spec_(28, 4, 128, torch.bfloat16), _caps(ns))),
                      "configs": [list(D().launch_config(block_seq=bs, max_context_len=n, requires_attention_scores=sc))
                                  for bs, n, sc in ((256, 4224, True), (64, 100, False), (1024, 131072, False))],
                      "base_raises": _err(lambda: ns.DecodeAttentionLaunchProvider().launch_config(
                          block_seq=1, max_context_len=1, requires_attention_scores=False))["err"]}
    # on a device no specialised provider was profiled for, the default answers (whatever else is registered declines)
    spec = Spec_(28, 4, 128, torch.bfloat16)
    generic = _caps(ns, platform=ns.PlatformEnum.CPU, device_type="cpu", device_name="generic", compute_capability=None)
    res = ns.OpResolver(ns.DECODE_ATTENTION_LAUNCH_REGISTRY).resolve(spec, generic)
    out["generic_device"] = {"chosen": res.provider.name, "n_rejected": len(res.rejected),
                             "rejected_have_reasons": all(isinstance(r, str) and r for _, r in res.rejected)}
    op = ns.PreparedDecodeAttentionLaunchOp(spec, res.provider)
    out["prepared"] = {"name": op.name, "spec_is_kept": op.spec is spec, "provider_is_kept": op.provider is res.provider,
                       "config": list(op.launch_config(block_seq=256, max_context_len=4224, requires_attention_scores=True))}
    # prepare_* on the CPU platform
    saved = os.environ.get("SPARSEVLLM_PLATFORM")
    os.environ["SPARSEVLLM_PLATFORM"] = "cpu"
    ns.platforms._set_current_platform_for_tests(None)
    try:
        op = ns.prepare_decode_attention_launch_op(spec, device_index=0)
        op2 = ns.prepare_decode_attention_launch_op(spec)
        out["prepare_on_cpu"] = {"name": op.name, "name_default_index": op2.name,
                                 "config": list(op.launch_config(block_seq=512, max_context_len=9, requires_attention_scores=False))}
    finally:
        ns.platforms._set_current_platform_for_tests(None)
        os.environ.pop("SPARSEVLLM_PLATFORM", None)
        if saved is not None:
            os.environ["SPARSEVLLM_PLATFORM"] = saved
    return out


def scenario_platforms(ns):
    import torch
    P = ns.platforms
    saved = os.environ.get("SPARSEVLLM_PLATFORM")
    out = {}
    try:
        os.environ["SPARSEVLLM_PLATFORM"] = "cpu"
        P._set_current_platform_for_tests(None)
        p = P.get_current_platform()
        caps = p.get_device_caps()
        out["cpu"] = {"name": p.name, "device": str(p.get_device(3)), "is_cpu_device": p.get_device(3) == torch.device("cpu"),
                      "backend": p.get_distributed_backend(), "graph": p.supports_graph_capture(),
                      "caps": {"platform": caps.platform.name, "device_type": caps.device_type,
                               "bf16": caps.supports_bfloat16, "fp8": caps.supports_native_fp8},
                      "supports_inference": p.supports_inference(), "validate_inference": _err(p.validate_inference),
                      "kind": [p.is_cpu(), p.is_cuda(), p.is_rocm(), p.is_npu(), p.is_cuda_alike()],
                      "same_object_again": P.get_current_platform() is p, "attr_access": P.current_platform is p,
                      "dispatch_key": p.get_dispatch_key(),
                      "validate_config_graph": _err(lambda: p.validate_config(type("C", (), {"decode_cuda_graph": True})()))}
        os.environ["SPARSEVLLM_PLATFORM"] = "missing_test_platform"
        P._set_current_platform_for_tests(None)
        r = _err(lambda: P.get_current_platform().name)
        out["unknown"] = {"err": r.get("err"), "mentions_env": "SPARSEVLLM_PLATFORM" in r.get("msg", ""),
                          "mentions_value": "missing_test_platform" in r.get("msg", "")}
        out["module_attr_error"] = _err(lambda: P.no_such_attribute)["err"]
    finally:
        P._set_current_platform_for_tests(None)
        os.environ.pop("SPARSEVLLM_PLATFORM", None)
        if saved is not None:
            os.environ["SPARSEVLLM_PLATFORM"] = saved
    return out


SCENARIOS = {"registry": scenario_registry, "runtime_stats": scenario_runtime_stats, "version": scenario_version,
             "types": scenario_types, "decode_launch": scenario_decode_launch, "platforms": scenario_platforms}


def run_all(ns):
    return {name: fn(ns) for name, fn in SCENARIOS.items()}
