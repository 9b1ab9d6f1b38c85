"""Import-path package of the MI355X build under the reference's dotted names.

The reference's drivers do `from src.modeling.alpro_models import ...`, `from src.utils.load_save import ...` AND import many
modules this repo does not rebuild (`src.datasets.*`, `src.configs.*`, `src.optimization.*`, `src.utils.basic_utils`, ...).  This
package therefore EXTENDS instead of shadowing: with the reference checkout on sys.path after this repo, every `src` directory on the
path contributes to the package; the hot-path modules defined here win (they come first), everything else resolves to the
reference's own file.  (pkgutil.extend_path; the reference's `src/modeling` and `src/utils` have no __init__.py and join as
namespace portions.)"""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
