"""reference vila/conversation.py: the HALVA scripts only use the `v1` (vicuna_v1) template, shared with the llava
package.  The module object itself is aliased so `conversation_lib.default_conversation = ...` is seen by both."""
import sys

from llava import conversation as _conv

sys.modules[__name__] = _conv
