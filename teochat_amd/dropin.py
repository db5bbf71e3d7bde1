"""Make `import videollava...` resolve to this package for the hot path.

    import teochat_amd.dropin; teochat_amd.dropin.install()
    from videollava.eval.eval import load_model
    from videollava.eval.inference import run_inference_single

Only the modules on the path are aliased; anything else under `videollava` raises ImportError.
"""
import sys
import types


def install():
    from . import builder, constants, conversation, detection, eval as teo_eval, inference, metrics, mm_utils, model
    pkg = types.ModuleType("videollava")
    pkg.__path__ = []
    pkg.__teo_dropin__ = True
    evalpkg = types.ModuleType("videollava.eval")
    evalpkg.__path__ = []
    modelpkg = types.ModuleType("videollava.model")
    modelpkg.__path__ = []
    modelpkg.LlavaLlamaForCausalLM = model.LlavaLlamaForCausalLM
    from .config import LlavaConfig
    modelpkg.LlavaConfig = LlavaConfig
    mods = {"videollava": pkg, "videollava.constants": constants, "videollava.conversation": conversation,
            "videollava.mm_utils": mm_utils, "videollava.eval": evalpkg, "videollava.eval.eval": teo_eval,
            "videollava.eval.inference": inference, "videollava.eval.classification": metrics, "videollava.eval.detection": detection,
            "videollava.model": modelpkg,
            "videollava.model.builder": builder}
    sys.modules.update(mods)
    pkg.constants, pkg.conversation, pkg.mm_utils, pkg.eval, pkg.model = constants, conversation, mm_utils, evalpkg, modelpkg
    evalpkg.eval, evalpkg.inference, evalpkg.classification, evalpkg.detection = teo_eval, inference, metrics, detection
    modelpkg.builder = builder
    return pkg
