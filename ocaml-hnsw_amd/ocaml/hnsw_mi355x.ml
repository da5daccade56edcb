(* hnsw_mi355x.ml -- OCaml side of the drop-in: ctypes-foreign binding of libhnsw_mi355x.so
   (include/hnsw_mi355x.h) plus the flatten shims that turn the reference's own graph containers
   into the tables the C ABI takes.

   SOURCE ONLY in this repository: the build image has no OCaml toolchain (no ocaml / opam / dune),
   so this file is not compiled or tested here; the identical ABI is exercised from Python ctypes
   (tests/) and C++ (host/hnsw_front.hpp).  It is written against the reference's modules as they
   are: Ohnsw (lib/ohnsw.ml) and Hnsw.Ba (lib/hnsw.ml:817-819).

   What stays OCaml: the graph builder (Ohnsw.build_batch_bigarray, Hnsw.Ba.build) and every
   signature.  What moves: the bodies of knn / knn_batch*, i.e. the search_one + search_k loops. *)
open Ctypes
open Foreign

let lib = Dl.dlopen ~filename:"libhnsw_mi355x.so" ~flags:[ Dl.RTLD_NOW ]

(* ---- C structs ---------------------------------------------------------------------------- *)
type layer_desc
let layer_desc : layer_desc structure typ = structure "hnsw_layer_desc"
let ld_n_nodes = field layer_desc "n_nodes" int64_t
let ld_nodes = field layer_desc "nodes" (ptr int64_t)
let ld_deg = field layer_desc "deg" (ptr int32_t)
let ld_nbr = field layer_desc "nbr" (ptr int32_t)
let () = seal layer_desc

type index_desc
let index_desc : index_desc structure typ = structure "hnsw_index_desc"
let d_vectors = field index_desc "vectors" (ptr float)
let d_n = field index_desc "n" int64_t
let d_d = field index_desc "d" int32_t
let d_row_stride = field index_desc "row_stride" int64_t
let d_metric = field index_desc "metric" int32_t
let d_id_base = field index_desc "id_base" int32_t
let d_max_degree0 = field index_desc "max_degree0" int32_t
let d_max_degree = field index_desc "max_degree" int32_t
let d_max_layer = field index_desc "max_layer" int32_t
let d_entry_point = field index_desc "entry_point" int64_t
let d_deg0 = field index_desc "deg0" (ptr int32_t)
let d_nbr0 = field index_desc "nbr0" (ptr int32_t)
let d_upper = field index_desc "upper" (ptr layer_desc)
let d_expected_ef = field index_desc "expected_ef" int32_t
let d_expected_semantics = field index_desc "expected_semantics" int32_t
let () = seal index_desc

type search_params
let search_params : search_params structure typ = structure "hnsw_search_params"
let p_ef = field search_params "ef" int32_t
let p_k = field search_params "k" int32_t
let p_fill = field search_params "fill" int32_t
let p_semantics = field search_params "semantics" int32_t
let () = seal search_params

type index = unit ptr
let index : index typ = ptr void

(* ---- entry points (the OCaml 4.x runtime lock is released around the blocking calls) -------- *)
let hnsw_last_error = foreign ~from:lib "hnsw_last_error" (void @-> returning string)
let hnsw_index_create =
  foreign ~from:lib ~release_runtime_lock:true "hnsw_index_create"
    (ptr index_desc @-> int32_t @-> ptr index @-> returning int32_t)
let hnsw_index_destroy = foreign ~from:lib "hnsw_index_destroy" (index @-> returning int32_t)
let hnsw_search_batch =
  foreign ~from:lib ~release_runtime_lock:true "hnsw_search_batch"
    (index @-> ptr float @-> int64_t @-> int64_t @-> ptr search_params @-> ptr int32_t @-> ptr float
     @-> ptr uint32_t @-> ptr uint32_t @-> returning int32_t)
let hnsw_search_layer_batch =
  foreign ~from:lib ~release_runtime_lock:true "hnsw_search_layer_batch"
    (index @-> int32_t @-> ptr float @-> int64_t @-> int64_t @-> ptr int64_t @-> int32_t
     @-> ptr search_params @-> ptr int32_t @-> ptr float @-> ptr int32_t @-> ptr uint32_t
     @-> ptr uint32_t @-> returning int32_t)
let hnsw_search_one_batch =
  foreign ~from:lib ~release_runtime_lock:true "hnsw_search_one_batch"
    (index @-> int32_t @-> ptr float @-> int64_t @-> int64_t @-> ptr int64_t @-> ptr int64_t
     @-> ptr float @-> returning int32_t)
let hnsw_index_set_option =
  foreign ~from:lib "hnsw_index_set_option" (index @-> string @-> int64_t @-> returning int32_t)
(* bytes of one vector as the knn searches read it: d when the library serves byte-valued data (SIFT: float32
   values that are all integers 0..255) from its lossless byte copy of the rows, 4 d otherwise.  Nothing to do on
   this side: the copy is built by hnsw_index_create itself and is invisible to knn_batch* (same results, bit for
   bit); `hnsw_index_set_option idx "byte_rows" 0L` reads the float32 rows again. *)
let hnsw_index_row_bytes =
  foreign ~from:lib "hnsw_index_row_bytes" (index @-> ptr int64_t @-> returning int32_t)
let hnsw_index_kernel_times =
  foreign ~from:lib ~release_runtime_lock:true "hnsw_index_kernel_times"
    (index @-> ptr double @-> ptr double @-> ptr int32_t @-> returning int32_t)
type request = unit ptr
let request : request typ = ptr void
let hnsw_search_submit =
  foreign ~from:lib ~release_runtime_lock:true "hnsw_search_submit"
    (index @-> ptr float @-> int64_t @-> int64_t @-> ptr search_params @-> ptr request @-> returning int32_t)
let hnsw_search_wait =
  foreign ~from:lib ~release_runtime_lock:true "hnsw_search_wait"
    (request @-> ptr int32_t @-> ptr float @-> ptr uint32_t @-> ptr uint32_t @-> returning int32_t)
type multi = unit ptr
let multi : multi typ = ptr void
let hnsw_multi_create =
  foreign ~from:lib ~release_runtime_lock:true "hnsw_multi_create"
    (ptr index_desc @-> ptr int32_t @-> int32_t @-> ptr multi @-> returning int32_t)
let hnsw_multi_destroy = foreign ~from:lib "hnsw_multi_destroy" (multi @-> returning int32_t)
let hnsw_multi_search_batch =
  foreign ~from:lib ~release_runtime_lock:true "hnsw_multi_search_batch"
    (multi @-> ptr float @-> int64_t @-> int64_t @-> ptr search_params @-> ptr int32_t @-> ptr float
     @-> ptr uint32_t @-> ptr uint32_t @-> returning int32_t)
(* sharded search + RCCL all-gather, results left on every device: d_ids / d_dist are arrays of
   n_devices device pointers written by the library *)
let hnsw_multi_search_batch_device =
  foreign ~from:lib ~release_runtime_lock:true "hnsw_multi_search_batch_device"
    (multi @-> ptr float @-> int64_t @-> int64_t @-> ptr search_params @-> ptr (ptr int32_t) @-> ptr (ptr float)
     @-> returning int32_t)
let hnsw_multi_debug_counters =
  foreign ~from:lib "hnsw_multi_debug_counters" (multi @-> ptr int64_t @-> returning int32_t)
let hnsw_multi_copy_result =
  foreign ~from:lib ~release_runtime_lock:true "hnsw_multi_copy_result"
    (multi @-> int32_t @-> ptr int32_t @-> ptr float @-> returning int32_t)

(* the rest of include/hnsw_mi355x.h: single query, gathered distances, select_neighbours, the device builder,
   save / load, per-layer statistics, host-array registration *)
let hnsw_knn =
  foreign ~from:lib ~release_runtime_lock:true "hnsw_knn"
    (index @-> ptr float @-> ptr search_params @-> ptr int32_t @-> ptr float @-> ptr int32_t @-> returning int32_t)
let hnsw_distance_batch =
  foreign ~from:lib ~release_runtime_lock:true "hnsw_distance_batch"
    (index @-> ptr float @-> int64_t @-> int64_t @-> ptr int32_t @-> int32_t @-> ptr float @-> returning int32_t)
let hnsw_select_neighbours_batch =
  foreign ~from:lib ~release_runtime_lock:true "hnsw_select_neighbours_batch"
    (index @-> ptr float @-> int64_t @-> int64_t @-> ptr int32_t @-> ptr int32_t @-> int32_t @-> int32_t @-> int32_t
     @-> ptr int32_t @-> ptr int32_t @-> ptr int32_t @-> returning int32_t)
type build_params
let build_params : build_params structure typ = structure "hnsw_build_params"
let b_num_connections = field build_params "num_connections" int32_t
let b_efc = field build_params "num_nodes_search_construction" int32_t
let b_metric = field build_params "metric" int32_t
let b_id_base = field build_params "id_base" int32_t
let b_seed = field build_params "seed" uint64_t
let b_max_batch = field build_params "max_batch" int32_t
let b_batch_div = field build_params "batch_div" int32_t
let b_expected_ef = field build_params "expected_ef" int32_t
let b_expected_semantics = field build_params "expected_semantics" int32_t
let () = seal build_params
let hnsw_build =
  foreign ~from:lib ~release_runtime_lock:true "hnsw_build"
    (ptr float @-> int64_t @-> int32_t @-> int64_t @-> ptr build_params @-> int32_t @-> ptr index @-> returning int32_t)
let hnsw_index_save = foreign ~from:lib ~release_runtime_lock:true "hnsw_index_save" (index @-> string @-> returning int32_t)
let hnsw_index_load =
  foreign ~from:lib ~release_runtime_lock:true "hnsw_index_load" (string @-> int32_t @-> ptr index @-> returning int32_t)
type layer_stats
let layer_stats : layer_stats structure typ = structure "hnsw_layer_stats"
let ls_num_nodes = field layer_stats "num_nodes" int64_t
let ls_min_degree = field layer_stats "min_degree" int32_t
let ls_max_degree = field layer_stats "max_degree" int32_t
let ls_mean_degree = field layer_stats "mean_degree" double
let ls_num_isolated = field layer_stats "num_isolated" int64_t
let () = seal layer_stats
let hnsw_index_layer_stats =
  foreign ~from:lib "hnsw_index_layer_stats" (index @-> int32_t @-> ptr layer_stats @-> returning int32_t)
let hnsw_index_layer_isolated =
  foreign ~from:lib "hnsw_index_layer_isolated"
    (index @-> int32_t @-> ptr int64_t @-> int64_t @-> ptr int64_t @-> returning int32_t)
let hnsw_index_locality_codes =
  foreign ~from:lib ~release_runtime_lock:true "hnsw_index_locality_codes" (index @-> ptr int32_t @-> returning int32_t)
let hnsw_index_visited_blocks =
  foreign ~from:lib ~release_runtime_lock:true "hnsw_index_visited_blocks" (index @-> ptr search_params @-> ptr int32_t @-> returning int32_t)
let hnsw_index_prepare =
  foreign ~from:lib ~release_runtime_lock:true "hnsw_index_prepare" (index @-> ptr search_params @-> returning int32_t)
let hnsw_abi_version = foreign ~from:lib "hnsw_abi_version" (void @-> returning int32_t)
(* HNSW_ABI_VERSION of include/hnsw_mi355x.h this binding was written against: an older or newer library is refused at load
   time (version 2: hnsw_search_batch_h2d, hnsw_host_alloc / hnsw_host_free, hnsw_index_layer_isolated, hnsw_multi_debug_counters,
   hnsw_index_locality_codes; hnsw_index_info.row_format; the empty-layer values of hnsw_index_layer_stats; version 3:
   expected_ef / expected_semantics at the end of hnsw_index_desc and hnsw_build_params, hnsw_index_prepare, index file format 2) *)
let expected_abi_version = 3l
let () =
  let v = hnsw_abi_version () in
  if v <> expected_abi_version then
    failwith (Printf.sprintf "libhnsw_mi355x.so speaks ABI version %ld, this binding %ld" v expected_abi_version)
let hnsw_device_count = foreign ~from:lib "hnsw_device_count" (ptr int32_t @-> returning int32_t)
type index_info
let index_info : index_info structure typ = structure "hnsw_index_info"
let ii_n = field index_info "n" int64_t
let ii_d = field index_info "d" int32_t
let ii_metric = field index_info "metric" int32_t
let ii_id_base = field index_info "id_base" int32_t
let ii_max_degree0 = field index_info "max_degree0" int32_t
let ii_max_degree = field index_info "max_degree" int32_t
let ii_max_layer = field index_info "max_layer" int32_t
let ii_entry_point = field index_info "entry_point" int64_t
let ii_device_bytes = field index_info "device_bytes" int64_t
let ii_row_stride_bytes = field index_info "row_stride_bytes" int64_t
let ii_device = field index_info "device" int32_t
let ii_row_format = field index_info "row_format" int32_t
let () = seal index_info
let hnsw_index_get_info = foreign ~from:lib "hnsw_index_get_info" (index @-> ptr index_info @-> returning int32_t)
(* the flattened graph of a device index (built there by hnsw_build, or loaded from a file) back to the host *)
let hnsw_index_export_layer0 =
  foreign ~from:lib ~release_runtime_lock:true "hnsw_index_export_layer0" (index @-> ptr int32_t @-> ptr int32_t @-> returning int32_t)
let hnsw_index_export_upper_count =
  foreign ~from:lib "hnsw_index_export_upper_count" (index @-> int32_t @-> ptr int64_t @-> returning int32_t)
let hnsw_index_export_upper =
  foreign ~from:lib ~release_runtime_lock:true "hnsw_index_export_upper"
    (index @-> int32_t @-> ptr int64_t @-> ptr int32_t @-> ptr int32_t @-> returning int32_t)
let hnsw_multi_num_replicas = foreign ~from:lib "hnsw_multi_num_replicas" (multi @-> ptr int32_t @-> returning int32_t)
let hnsw_multi_replica = foreign ~from:lib "hnsw_multi_replica" (multi @-> int32_t @-> ptr index @-> returning int32_t)
(* device-pointer entry points (hnsw_search_batch_device, hnsw_distance_batch_device): for callers that already hold
   HIP device pointers and a stream; an OCaml program has neither, so they are bound as raw pointers only *)
let hnsw_search_batch_device =
  foreign ~from:lib "hnsw_search_batch_device"
    (index @-> ptr float @-> int64_t @-> int64_t @-> ptr search_params @-> ptr int32_t @-> ptr float @-> ptr uint32_t
     @-> ptr uint32_t @-> ptr uint32_t @-> ptr void @-> returning int32_t)
let hnsw_search_batch_h2d =
  foreign ~from:lib "hnsw_search_batch_h2d"
    (index @-> ptr float @-> int64_t @-> int64_t @-> ptr search_params @-> ptr int32_t @-> ptr float @-> ptr uint32_t
     @-> ptr uint32_t @-> ptr uint32_t @-> ptr void @-> returning int32_t)
let hnsw_distance_batch_device =
  foreign ~from:lib "hnsw_distance_batch_device"
    (index @-> ptr float @-> int64_t @-> int64_t @-> ptr int32_t @-> int32_t @-> ptr float @-> ptr void @-> returning int32_t)
let hnsw_host_register = foreign ~from:lib "hnsw_host_register" (ptr void @-> int64_t @-> returning int32_t)
let hnsw_host_unregister = foreign ~from:lib "hnsw_host_unregister" (ptr void @-> returning int32_t)
let hnsw_host_alloc = foreign ~from:lib "hnsw_host_alloc" (ptr (ptr void) @-> int64_t @-> returning int32_t)
let hnsw_host_free = foreign ~from:lib "hnsw_host_free" (ptr void @-> returning int32_t)

(* Error convention -> the reference's exceptions (lib/ohnsw.ml:25,343,862) *)
let check rc =
  match Int32.to_int rc with
  | 0 -> ()
  | -1 | -2 | -3 -> invalid_arg (hnsw_last_error ())   (* BAD_ARG | EMPTY_INDEX | DEGREE_OVERFLOW *)
  | _ -> failwith (hnsw_last_error ())

(* ---- flatten ------------------------------------------------------------------------------- *)
module A1 = Bigarray.Array1
module A2 = Bigarray.Array2

type flat = {
  deg0 : (int32, Bigarray.int32_elt, Bigarray.c_layout) A1.t;
  nbr0 : (int32, Bigarray.int32_elt, Bigarray.c_layout) A2.t;   (* n x width0, iteration order *)
  upper : ((int64, Bigarray.int64_elt, Bigarray.c_layout) A1.t
           * (int32, Bigarray.int32_elt, Bigarray.c_layout) A1.t
           * (int32, Bigarray.int32_elt, Bigarray.c_layout) A2.t) array;
  entry_point : int;
  max_layer : int;
  width0 : int;          (* row width of layer 0: 2M, or the longest list found *)
  width_upper : int;     (* row width of the layers above: M, or the longest list found *)
}

(* Ohnsw.Hgraph.t (lib/ohnsw.ml:307-312): every layer is dense over all n nodes (:328-330);
   Graph.iter_neighbours (:176-180) walks nodes, Neighbours.iter (:127) walks a list head first --
   exactly the order search_k folds over (:570).  A list longer than its row raises: the C side
   would refuse it too (HNSW_ERR_DEGREE_OVERFLOW); nothing is ever truncated. *)
(* The row widths: with ~num_connections:m they are the caps the builder guarantees (2m on layer 0, m
   above, lib/ohnsw.ml:818-828); without it -- an Ohnsw.Hgraph.t does not record M -- they are the
   longest list found on layer 0 / on the layers above (one pass over the lists). *)
let ohnsw_widths (h : _ Ohnsw.Hgraph.t) =
  let n = Ohnsw.Hgraph.num_nodes h in
  let longest g =
    let w = ref 1 in
    for i = 0 to n - 1 do
      w := max !w (Ohnsw.Neighbours.length (Ohnsw.Graph.adjacent g i))
    done;
    !w in
  let w0 = longest (Ohnsw.Hgraph.layer h 0) in
  let wu = ref 1 in
  for l = 1 to Ohnsw.Hgraph.max_layer h do wu := max !wu (longest (Ohnsw.Hgraph.layer h l)) done;
  w0, !wu

let flatten_ohnsw ?num_connections (h : _ Ohnsw.Hgraph.t) : flat =
  let n = Ohnsw.Hgraph.num_nodes h in
  let max_layer = Ohnsw.Hgraph.max_layer h in
  let width0, width_upper = match num_connections with
    | Some m -> 2 * m, m
    | None -> ohnsw_widths h in
  let row g node width (dst : (int32, _, _) A2.t) r =
    let nb = Ohnsw.Graph.adjacent g node in
    if Ohnsw.Neighbours.length nb > width then invalid_arg "flatten: degree exceeds row width";
    let j = ref 0 in
    Ohnsw.Neighbours.iter nb ~f:(fun e -> dst.{r, !j} <- Int32.of_int e; incr j);
    !j
  in
  let deg0 = A1.create Bigarray.int32 Bigarray.c_layout n in
  let nbr0 = A2.create Bigarray.int32 Bigarray.c_layout n width0 in
  A2.fill nbr0 (-1l);
  let g0 = Ohnsw.Hgraph.layer h 0 in
  for i = 0 to n - 1 do deg0.{i} <- Int32.of_int (row g0 i width0 nbr0 i) done;
  let upper =
    Array.init max_layer (fun l ->
        let g = Ohnsw.Hgraph.layer h (l + 1) in
        (* nodes present on the layer: those with links there, plus the entry point *)
        let present i =
          Ohnsw.Neighbours.length (Ohnsw.Graph.adjacent g i) > 0
          || Ohnsw.Hgraph.entry_point h = Some i in
        let ids = List.filter present (List.init n (fun i -> i)) in
        let c = List.length ids in
        let nodes = A1.create Bigarray.int64 Bigarray.c_layout c in
        let deg = A1.create Bigarray.int32 Bigarray.c_layout c in
        let nbr = A2.create Bigarray.int32 Bigarray.c_layout c width_upper in
        A2.fill nbr (-1l);
        List.iteri (fun s i ->
            nodes.{s} <- Int64.of_int i;
            deg.{s} <- Int32.of_int (row g i width_upper nbr s)) ids;
        (nodes, deg, nbr))
  in
  { deg0; nbr0; upper; max_layer; width0; width_upper;
    entry_point = (match Ohnsw.Hgraph.entry_point h with Some e -> e | None -> -1) }

(* Hnsw.Ba.Hgraph.t (lib/hnsw.ml:342-348): layers = Map int -> MapGraph.t, a MapGraph holding
   connections = Map int -> NeighbourList.t ({list; length}, lib/hnsw.ml:33-45).  Node ids are the
   1-based matrix columns (lib/hnsw.ml:325); a node without an entry in a layer's map has no
   neighbours there (MapGraph.adjacent, lib/hnsw.ml:146-149).  NeighbourList.fold walks the list head
   first (lib/hnsw.ml:44-45), which is the order Search.search folds over (lib/hnsw_algo.ml:381-383).
   Pass the result to [create ~id_base:1]. *)
let ba_widths (h : Hnsw.Ba.Hgraph.t) =
  let module G = Hnsw.Ba.Hgraph in
  let longest (g : Hnsw.MapGraph.t) =
    Base.Map.fold g.Hnsw.MapGraph.connections ~init:1 ~f:(fun ~key:_ ~data w ->
        max w (Hnsw.NeighbourList.length data)) in
  let w0 = longest (G.layer h 0) in
  let wu = ref 1 in
  for l = 1 to G.max_layer h do wu := max !wu (longest (G.layer h l)) done;
  w0, !wu

let flatten_ba ?num_connections (h : Hnsw.Ba.Hgraph.t) : flat =
  let module G = Hnsw.Ba.Hgraph in
  let n = G.max_node_id h in                       (* = Values.length, lib/hnsw.ml:394 *)
  let max_layer = G.max_layer h in
  let width0, width_upper = match num_connections with
    | Some m -> 2 * m, m
    | None -> ba_widths h in
  let fill_row (nl : Hnsw.NeighbourList.t) width (dst : (int32, _, _) A2.t) r =
    if Hnsw.NeighbourList.length nl > width then invalid_arg "flatten: degree exceeds row width";
    let j = ref 0 in
    Hnsw.NeighbourList.fold nl ~init:() ~f:(fun () e -> dst.{r, !j} <- Int32.of_int e; incr j);
    !j
  in
  let deg0 = A1.create Bigarray.int32 Bigarray.c_layout n in
  let nbr0 = A2.create Bigarray.int32 Bigarray.c_layout n width0 in
  A1.fill deg0 0l; A2.fill nbr0 (-1l);
  let g0 = G.layer h 0 in
  Base.Map.iteri g0.Hnsw.MapGraph.connections ~f:(fun ~key ~data ->
      deg0.{key - 1} <- Int32.of_int (fill_row data width0 nbr0 (key - 1)));
  let upper =
    Array.init max_layer (fun l ->
        let g = G.layer h (l + 1) in
        let c = Base.Map.length g.Hnsw.MapGraph.connections in
        let nodes = A1.create Bigarray.int64 Bigarray.c_layout c in
        let deg = A1.create Bigarray.int32 Bigarray.c_layout c in
        let nbr = A2.create Bigarray.int32 Bigarray.c_layout c width_upper in
        A2.fill nbr (-1l);
        let s = ref 0 in
        Base.Map.iteri g.Hnsw.MapGraph.connections ~f:(fun ~key ~data ->
            nodes.{!s} <- Int64.of_int key;
            deg.{!s} <- Int32.of_int (fill_row data width_upper nbr !s);
            incr s);
        (nodes, deg, nbr))
  in
  { deg0; nbr0; upper; max_layer; width0; width_upper;
    entry_point = G.entry_point h (* -1 when empty, lib/hnsw.ml:391 *) }

(* The inverse for the functor path: rebuild a Hnsw.Ba.Hgraph.t (lib/hnsw.ml:342-348) over the value
   matrix from flattened tables with 1-based ids.  The record fields are the reference's own: layers is a
   Map from layer to MapGraph.t = { connections : NeighbourList.t Map.M(Int).t; max_node_id }
   (lib/hnsw.ml:122-135), a NeighbourList is { list; length } with the list in fold order (head first,
   lib/hnsw.ml:33-45) -- the row order.  Nodes without links on a layer get no map entry, as
   MapGraph.adjacent expects (lib/hnsw.ml:146-149); the tables hold symmetric links already. *)
let unflatten_ba (values : Lacaml.S.mat) (f : flat) : Hnsw.Ba.Hgraph.t =
  let n = Lacaml.S.Mat.dim2 values in
  let list_of_row (nbr : (int32, _, _) A2.t) r deg =
    let l = ref [] in
    for j = deg - 1 downto 0 do l := Int32.to_int nbr.{r, j} :: !l done;
    { Hnsw.NeighbourList.list = !l; length = deg } in
  let graph_of rows =
    { Hnsw.MapGraph.connections =
        List.fold_left (fun m (node, nl) -> Base.Map.set m ~key:node ~data:nl)
          (Base.Map.empty (module Base.Int)) rows;
      max_node_id = n } in
  let layer0 =
    let rows = ref [] in
    for i = A1.dim f.deg0 - 1 downto 0 do
      let deg = Int32.to_int f.deg0.{i} in
      if deg > 0 then rows := (i + 1, list_of_row f.nbr0 i deg) :: !rows
    done;
    graph_of !rows in
  let layers = ref (Base.Map.set (Base.Map.empty (module Base.Int)) ~key:0 ~data:layer0) in
  Array.iteri (fun l (nodes, deg, nbr) ->
      let rows = ref [] in
      for r = A1.dim nodes - 1 downto 0 do
        rows := (Int64.to_int nodes.{r}, list_of_row nbr r (Int32.to_int deg.{r})) :: !rows
      done;
      layers := Base.Map.set !layers ~key:(l + 1) ~data:(graph_of !rows)) f.upper;
  { Hnsw.Ba.Hgraph.layers = !layers; max_layer = f.max_layer; entry_point = f.entry_point; values }

(* The inverse: rebuild an Ohnsw.Hgraph.t from the flattened tables (e.g. an index built on the device by
   hnsw_build and fetched with hnsw_index_export_*), so that the OCaml builder can keep inserting into
   it.  Every Neighbours.t gets its list in the stored order (Neighbours.iter order = search_k's
   fold order, :127,:570): Neighbours.add conses (:116-118), so a row is pushed back to front.  The
   tables hold symmetric links already, so the rows are written with Vector.set, not with
   Graph.set_connections (which would add every link twice, :182-196). *)
let unflatten_ohnsw (distance : 'a Ohnsw.distance) (value : 'a Ohnsw.value) (f : flat) : 'a Ohnsw.Hgraph.t =
  let h = Ohnsw.Hgraph.create distance value in
  let n = A1.dim f.deg0 in
  for _ = 1 to n do ignore (Ohnsw.Hgraph.add_node h) done;          (* lib/ohnsw.ml:328-330 *)
  Ohnsw.Hgraph.set_max_layer h f.max_layer;                            (* :347-351 *)
  let neighbours_of_row (nbr : (int32, _, _) A2.t) r deg =
    let nb = Ohnsw.Neighbours.create () in
    for j = deg - 1 downto 0 do Ohnsw.Neighbours.add nb (Int32.to_int nbr.{r, j}) done;
    nb in
  let g0 = Ohnsw.Hgraph.layer h 0 in
  for i = 0 to n - 1 do
    Ohnsw.Vector.set g0 i (neighbours_of_row f.nbr0 i (Int32.to_int f.deg0.{i}))
  done;
  Array.iteri (fun l (nodes, deg, nbr) ->
      let g = Ohnsw.Hgraph.layer h (l + 1) in
      for r = 0 to A1.dim nodes - 1 do
        Ohnsw.Vector.set g (Int64.to_int nodes.{r}) (neighbours_of_row nbr r (Int32.to_int deg.{r}))
      done) f.upper;
  if f.entry_point >= 0 then Ohnsw.Hgraph.set_entry_point h f.entry_point;   (* :341-344 *)
  h

(* ---- device-resident index ------------------------------------------------------------------ *)
(* [scratch]: page-locked result matrices per (k, nq) shape, see [scratch_for]: they belong to the HANDLE, so two threads that
   search two different indices (one thread at a time per handle, INTEGRATION.md) never write into the same matrices *)
type result_scratch = (int * int, (int32, Bigarray.int32_elt, Bigarray.fortran_layout) A2.t * Lacaml.S.mat) Hashtbl.t
type t = { handle : index; k_base : int; dim : int; scratch : result_scratch }

(* ?expected_ef (and ?expected_semantics: 0 Ohnsw's accept rule, 1 the functor's): the upload also does, once, what the first
   search with these parameters would otherwise do inside the call (hnsw_index_prepare) *)
let create ?(device = 0) ?(metric = 0) ?(expected_ef = 0) ?(expected_semantics = 0) ~id_base (vectors : Lacaml.S.mat) (f : flat) : t =
  (* a Lacaml.S.mat is a Fortran-layout dim x n Bigarray: in memory, n rows of dim floats *)
  let dim = A2.dim1 vectors and n = A2.dim2 vectors in
  let layers = CArray.make layer_desc (max 1 f.max_layer) in
  Array.iteri (fun l (nodes, deg, nbr) ->
      let ld = CArray.get layers l in
      setf ld ld_n_nodes (Int64.of_int (A1.dim nodes));
      setf ld ld_nodes (bigarray_start array1 nodes);
      setf ld ld_deg (bigarray_start array1 deg);
      setf ld ld_nbr (bigarray_start array2 nbr)) f.upper;
  let d = make index_desc in
  setf d d_vectors (bigarray_start array2 vectors);
  setf d d_n (Int64.of_int n); setf d d_d (Int32.of_int dim);
  setf d d_row_stride (Int64.of_int dim);
  setf d d_metric (Int32.of_int metric); setf d d_id_base (Int32.of_int id_base);
  setf d d_max_degree0 (Int32.of_int f.width0); setf d d_max_degree (Int32.of_int f.width_upper);
  setf d d_max_layer (Int32.of_int f.max_layer);
  setf d d_entry_point (Int64.of_int f.entry_point);
  setf d d_deg0 (bigarray_start array1 f.deg0); setf d d_nbr0 (bigarray_start array2 f.nbr0);
  setf d d_upper (CArray.start layers);
  setf d d_expected_ef (Int32.of_int expected_ef); setf d d_expected_semantics (Int32.of_int expected_semantics);
  let out = allocate index null in
  check (hnsw_index_create (addr d) (Int32.of_int device) out);
  let t = { handle = !@out; k_base = id_base; dim; scratch = Hashtbl.create 4 } in
  Gc.finalise (fun t -> ignore (hnsw_index_destroy t.handle)) t;
  t

(* A Lacaml-shaped matrix (float32, Fortran layout, dim x n) in page-locked memory the LIBRARY allocates (hnsw_host_alloc):
   knn_batch* reads such a query matrix and writes such result matrices straight from the device, without copies.
   Freed by the finaliser of the returned Bigarray (keep using the memory only through the returned value). *)
let alloc_mat ~dim ~n : Lacaml.S.mat =
  let out = allocate (ptr void) null in
  check (hnsw_host_alloc out (Int64.of_int (4 * dim * (max n 1))));
  let p = !@out in
  let m = bigarray_of_ptr array2 (n, dim) Bigarray.float32 (from_voidp float p) in          (* n rows of dim floats in memory ... *)
  let m = Bigarray.Array2.change_layout m Bigarray.fortran_layout in                      (* ... = a dim x n Fortran matrix, as Lacaml's *)
  Gc.finalise (fun _ -> ignore (hnsw_host_free p)) m;
  m

(* the same for the int32 id table of a batch (k x nq Fortran = [nq][k] in memory) *)
let alloc_ids ~k ~nq : (int32, Bigarray.int32_elt, Bigarray.fortran_layout) A2.t =
  let out = allocate (ptr void) null in
  check (hnsw_host_alloc out (Int64.of_int (4 * k * (max nq 1))));
  let p = !@out in
  let m = bigarray_of_ptr array2 (nq, k) Bigarray.int32 (from_voidp int32_t p) in
  let m = Bigarray.Array2.change_layout m Bigarray.fortran_layout in
  Gc.finalise (fun _ -> ignore (hnsw_host_free p)) m;
  m

(* Page-locked result scratch, one pair per handle and (k, nq) shape seen (a benchmark or serving loop repeats its shape): the
   search kernel writes every query's results straight into it as the query finishes -- no download step -- and [search] hands
   fresh matrices to its caller, as the reference does (lib/ohnsw.ml:879-881), by two blits of 4 k nq bytes each.
   (Allocating page-locked memory costs far more than a batch takes, so the scratch is kept, not the results.)  The table is
   a field of the handle (round 4 kept ONE table for the process: two threads searching two indices with the same batch shape
   -- the call releases the runtime lock -- then had their kernels write into the same matrices); it dies with the handle,
   its blocks through their own finalisers (hnsw_host_free). *)
let scratch_for (t : t) ~k ~nq =
  match Hashtbl.find_opt t.scratch (k, nq) with
  | Some s -> s
  | None ->
    if Hashtbl.length t.scratch >= 8 then Hashtbl.reset t.scratch;   (* shapes keep changing: do not hoard *)
    let s = (alloc_ids ~k ~nq, alloc_mat ~dim:k ~n:nq) in
    Hashtbl.replace t.scratch (k, nq) s;
    s

let search ?(semantics = 0) t (batch : Lacaml.S.mat) ~ef ~k ~fill =
  let nq = A2.dim2 batch in
  (* results as the reference lays them out: k x nq Fortran = [nq][k] in memory; written by the device into the
     page-locked scratch, then copied into fresh matrices for the caller *)
  let ids_s, dist_s = scratch_for t ~k ~nq in
  let p = make search_params in
  setf p p_ef (Int32.of_int ef); setf p p_k (Int32.of_int k); setf p p_fill (Int32.of_int fill);
  setf p p_semantics (Int32.of_int semantics);
  check (hnsw_search_batch t.handle (bigarray_start array2 batch) (Int64.of_int nq)
           (Int64.of_int t.dim) (addr p) (bigarray_start array2 ids_s)
           (bigarray_start array2 dist_s) (from_voidp uint32_t null) (from_voidp uint32_t null));
  let distances = Lacaml.S.Mat.create k nq in
  let ids = A2.create Bigarray.int32 Bigarray.fortran_layout k nq in
  A2.blit dist_s distances;
  A2.blit ids_s ids;
  ids, distances

(* ---- from the reference's own index values, nothing else needed ----------------------------------- *)

(* of_ohnsw : Lacaml.S.vec Ohnsw.Hgraph.t -> t.  An Ohnsw.Hgraph.t (lib/ohnsw.ml:307-312) holds no
   matrix and no M: the vectors are gathered through its own [value : int -> Lacaml.S.vec] closure
   (for a graph from build_batch_bigarray that is [fun i -> Mat.col batch (i+1)], :842) into a fresh
   dim x n matrix, and the row widths are the longest lists found (ohnsw_widths). *)
let of_ohnsw ?device ?metric (h : Lacaml.S.vec Ohnsw.Hgraph.t) : t =
  let n = Ohnsw.Hgraph.num_nodes h in
  if n = 0 then invalid_arg "knn: empty hgraph";                        (* lib/ohnsw.ml:862 *)
  let value = Ohnsw.Hgraph.value h in
  let dim = Lacaml.S.Vec.dim (value 0) in
  let vectors = Lacaml.S.Mat.create dim n in
  for i = 0 to n - 1 do
    A1.blit (value i) (Lacaml.S.Mat.col vectors (i + 1))                (* node i <-> column i+1, :842 *)
  done;
  create ?device ?metric ~id_base:0 vectors (flatten_ohnsw h)

(* of_ba : Hnsw.Ba.Hgraph.t -> t.  The functor path's Hgraph carries its value matrix
   (values : BaValues.t = Lacaml.S.mat, lib/hnsw.ml:346 with :296-301); node ids are its columns (1-based). *)
let of_ba ?device ?metric (h : Hnsw.Ba.Hgraph.t) : t =
  if Hnsw.Ba.Hgraph.is_empty h then invalid_arg "knn: empty hgraph";
  create ?device ?metric ~id_base:1 h.Hnsw.Ba.Hgraph.values (flatten_ba h)

(* One device index per live hgraph: an association list of ephemerons keyed by the hgraph value itself
   (physical equality), so an index is destroyed (Gc.finalise in [create]) once its hgraph is collected.
   An Ohnsw.Hgraph.t is mutable -- the OCaml builder may keep inserting --, so an entry also records
   (number of nodes, entry point, max layer) and is rebuilt when they no longer match: every insert adds a
   node (lib/ohnsw.ml:766-772), so a changed graph always shows. *)
type cached = { index : t; stamp : int * int * int }
let cache : (Obj.t, cached) Ephemeron.K1.t list ref = ref []

let cached_index (key : Obj.t) ~(stamp : int * int * int) ~(make : unit -> t) : t =
  let live = List.filter (fun e -> Ephemeron.K1.check_key e) !cache in
  let hit = List.find_opt (fun e ->
      match Ephemeron.K1.get_key e with Some k -> k == key | None -> false) live in
  match hit with
  | Some e when (match Ephemeron.K1.get_data e with Some c -> c.stamp = stamp | None -> false) ->
    cache := live;
    (match Ephemeron.K1.get_data e with Some c -> c.index | None -> assert false)
  | _ ->
    let index = make () in
    let e = Ephemeron.K1.create () in
    Ephemeron.K1.set_key e key;
    Ephemeron.K1.set_data e { index; stamp };
    cache := e :: List.filter (fun e' -> match hit with Some h -> e' != h | None -> true) live;
    index

let index_of_ohnsw (h : Lacaml.S.vec Ohnsw.Hgraph.t) : t =
  let stamp = (Ohnsw.Hgraph.num_nodes h,
               (match Ohnsw.Hgraph.entry_point h with Some e -> e | None -> -1),
               Ohnsw.Hgraph.max_layer h) in
  cached_index (Obj.repr h) ~stamp ~make:(fun () -> of_ohnsw h)

let index_of_ba (h : Hnsw.Ba.Hgraph.t) : t =
  (* persistent structure: a modified graph is a different value, the stamp is only a cheap guard *)
  let stamp = (Hnsw.Ba.Hgraph.max_node_id h, Hnsw.Ba.Hgraph.entry_point h, Hnsw.Ba.Hgraph.max_layer h) in
  cached_index (Obj.repr h) ~stamp ~make:(fun () -> of_ba h)

(* ---- the drop-in bodies ---------------------------------------------------------------------- *)

(* on an explicit handle *)
let ohnsw_knn_batch_bigarray (t : t) ~k (batch : Lacaml.S.mat) =
  let ids32, distances = search t batch ~ef:k ~k ~fill:0 (* NaN / -1, :880-881 *) in
  let nq = A2.dim2 batch in
  let ids = Array.init nq (fun j -> Array.init k (fun i -> Int32.to_int ids32.{i + 1, j + 1})) in
  ids, distances

(* Ohnsw.knn_batch_bigarray (lib/ohnsw.ml:877-897) with its exact signature, at the value type the GPU
   path needs:
     Lacaml.S.vec Ohnsw.Hgraph.t -> k:int -> Lacaml.S.mat -> int array array * Lacaml.S.mat
   An empty batch returns empty results, as the reference's fold does; an empty hgraph raises
   Invalid_argument "knn: empty hgraph" (:862) as soon as there is a query. *)
let knn_batch_bigarray (hgraph : Lacaml.S.vec Ohnsw.Hgraph.t) ~k (batch : Lacaml.S.mat) :
  int array array * Lacaml.S.mat =
  if Lacaml.S.Mat.dim2 batch = 0 then [||], Lacaml.S.Mat.create k 0
  else ohnsw_knn_batch_bigarray (index_of_ohnsw hgraph) ~k batch

(* Hnsw.Ba.knn_batch : t -> Lacaml.S.mat -> num_neighbours_search:int -> num_neighbours:int
   -> Lacaml.S.mat (lib/hnsw.ml:769-777): distances only, +inf filled (:771).
   semantics 1 = Nearest.insert_distance's rule (lib/hnsw.ml:494-506: a neighbour tied with max(W) is
   Inserted and expanded, W keeps the incumbent) with (distance, id) order among equal keys; 2 = the same +
   Nearest.nearest_k's output (the k farthest of W when num_neighbours_search > num_neighbours,
   lib/hnsw.ml:522-525) for callers that want the reference's defect reproduced. *)
let ba_knn_batch ?(nearest_k_compat = false) (t : t) (batch : Lacaml.S.mat) ~num_neighbours_search ~num_neighbours =
  snd (search ~semantics:(if nearest_k_compat then 2 else 1) t batch ~ef:num_neighbours_search ~k:num_neighbours ~fill:1)

(* ... and with the reference's signature (the body of MakeBatch.knn_batch at Distance = EuclideanBa) *)
let knn_batch ?nearest_k_compat (hgraph : Hnsw.Ba.Hgraph.t) (batch : Lacaml.S.mat) ~num_neighbours_search ~num_neighbours :
  Lacaml.S.mat =
  if Lacaml.S.Mat.dim2 batch = 0 then Lacaml.S.Mat.create num_neighbours 0
  else ba_knn_batch ?nearest_k_compat (index_of_ba hgraph) batch ~num_neighbours_search ~num_neighbours

(* Hnsw.Make(Batch).knn_batch (lib/hnsw.ml:781-807): the generic front-end returns a Batch.Distances.t filled
   through Batch.Distances.set with one value_distance list per query.  A Batch whose values are
   Lacaml.S.vec (the only kind the device path can take: BATCH with type value = Lacaml.S.vec) is served
   by gathering the batch into a matrix, one device call, and feeding Distances.set row by row. *)
module MakeBatchGpu (Batch : Hnsw.BATCH with type value = Lacaml.S.vec) = struct
  let knn_batch (t : t) (batch : Batch.t) ~num_neighbours_search ~num_neighbours : Batch.Distances.t =
    let nq = Batch.length batch in
    let distances = Batch.Distances.create ~len_batch:nq ~num_neighbours in
    if nq > 0 then begin
      let m = Lacaml.S.Mat.create t.dim nq in
      ignore (Batch.fold batch ~init:1 ~f:(fun j (row : Lacaml.S.vec) ->
          A1.blit row (Lacaml.S.Mat.col m j); j + 1));
      let ids, dist = search ~semantics:1 t m ~ef:num_neighbours_search ~k:num_neighbours ~fill:1 in
      for j = 1 to nq do
        let l = ref [] in
        for i = num_neighbours downto 1 do
          let node = Int32.to_int ids.{i, j} in
          if node >= t.k_base then
            l := { Hnsw_algo.node; distance_to_target = dist.{i, j} } :: !l
        done;
        Batch.Distances.set distances j !l                     (* lib/hnsw.ml:803-804: i counts from 1 *)
      done
    end;
    distances
end

(* Ohnsw.search_k (lib/ohnsw.ml:543-588) on one layer for ONE target: the start MinQueue as a node
   list, the result MinQueue as an ascending (node, distance) list.  ~semantics:1 is
   Hnsw_algo.Search.search (lib/hnsw_algo.ml:350-391). *)
let search_k ?(semantics = 0) (t : t) ~layer ~(start_nodes : int list) (target : Lacaml.S.vec) ~k =
  let ns = List.length start_nodes in
  let st = CArray.of_list int64_t (List.map Int64.of_int start_nodes) in
  let ids = CArray.make int32_t k and dist = CArray.make float k in
  let cnt = allocate int32_t 0l in
  let p = make search_params in
  setf p p_ef (Int32.of_int k); setf p p_k (Int32.of_int k); setf p p_fill 0l;
  setf p p_semantics (Int32.of_int semantics);
  check (hnsw_search_layer_batch t.handle (Int32.of_int layer) (bigarray_start array1 target) 1L
           (Int64.of_int t.dim) (CArray.start st) (Int32.of_int ns) (addr p) (CArray.start ids)
           (CArray.start dist) cnt (from_voidp uint32_t null) (from_voidp uint32_t null));
  List.init (Int32.to_int !@cnt) (fun i -> Int32.to_int (CArray.get ids i), CArray.get dist i)

(* Ohnsw.search_one (lib/ohnsw.ml:492-512): the node the greedy walk on `layer` ends on *)
let search_one (t : t) ~layer ~start_node (target : Lacaml.S.vec) =
  let st = allocate int64_t (Int64.of_int start_node) and node = allocate int64_t 0L in
  check (hnsw_search_one_batch t.handle (Int32.of_int layer) (bigarray_start array1 target) 1L
           (Int64.of_int t.dim) st node (from_voidp float null));
  Int64.to_int !@node

(* Batches in flight: [submit] copies the batch in and starts the search, [wait] returns what
   [search] would have.  A caller with more than one batch overlaps them
   (let r1 = submit t b1 ... in let r2 = submit t b2 ... in wait r1; wait r2): the next batch fills
   the drain of the previous one (round 3, SIFT1M-shaped workload, registered matrices, PCIe copies included:
   36 M q/s with two requests in flight against 16.7 M q/s for back-to-back synchronous calls). *)
type pending = { req : request; p_k : int; p_nq : int; keep : Lacaml.S.mat }

let submit ?(semantics = 0) t (batch : Lacaml.S.mat) ~ef ~k ~fill : pending =
  let nq = A2.dim2 batch in
  let p = make search_params in
  setf p p_ef (Int32.of_int ef); setf p p_k (Int32.of_int k); setf p p_fill (Int32.of_int fill);
  setf p p_semantics (Int32.of_int semantics);
  let out = allocate request null in
  check (hnsw_search_submit t.handle (bigarray_start array2 batch) (Int64.of_int nq)
           (Int64.of_int t.dim) (addr p) out);
  { req = !@out; p_k = k; p_nq = nq; keep = batch }

let wait (r : pending) =
  let distances = Lacaml.S.Mat.create r.p_k r.p_nq in
  let ids = A2.create Bigarray.int32 Bigarray.fortran_layout r.p_k r.p_nq in
  check (hnsw_search_wait r.req (bigarray_start array2 ids) (bigarray_start array2 distances)
           (from_voidp uint32_t null) (from_voidp uint32_t null));
  ids, distances

(* ---- single query ------------------------------------------------------------------------------ *)

(* one query through hnsw_knn: the (node, distance) pairs found, nearest first *)
let knn_on ?(semantics = 0) (t : t) (target : Lacaml.S.vec) ~ef ~k ~fill : (int * float) list =
  let ids = CArray.make int32_t k and dist = CArray.make float k in
  let cnt = allocate int32_t 0l in
  let p = make search_params in
  setf p p_ef (Int32.of_int ef); setf p p_k (Int32.of_int k); setf p p_fill (Int32.of_int fill);
  setf p p_semantics (Int32.of_int semantics);
  check (hnsw_knn t.handle (bigarray_start array1 target) (addr p) (CArray.start ids) (CArray.start dist) cnt);
  List.init (Int32.to_int !@cnt) (fun i -> Int32.to_int (CArray.get ids i), CArray.get dist i)

(* Ohnsw.knn (lib/ohnsw.ml:859-875).  The reference's signature is
     'a Hgraph.t -> Visited.t -> k:int -> 'a -> 'a MinQueue.t
   whose result type is abstract outside lib/ohnsw.ml (:404-416; test/test.ml:122-125 no longer type-checks against
   it): what a caller can do with it is pop it nearest first (:886-893), so that list is what is returned, and the
   scratch Visited.t (device-side state here) is not taken.  An empty hgraph raises Invalid_argument
   "knn: empty hgraph" (:861-862). *)
let knn (hgraph : Lacaml.S.vec Ohnsw.Hgraph.t) ~k (target : Lacaml.S.vec) : (int * float) list =
  knn_on (index_of_ohnsw hgraph) target ~ef:k ~k ~fill:0

(* Hnsw.Ba.knn (lib/hnsw.ml:763-767):
     t -> Lacaml.S.vec -> num_neighbours_search:int -> num_neighbours:int -> int Hnsw_algo.value_distance list
   1-based node ids, the functor path's accept rule (semantics 1); ~nearest_k_compat:true reproduces
   Nearest.nearest_k's output (lib/hnsw.ml:522-525) as in [knn_batch]. *)
let ba_knn ?(nearest_k_compat = false) (hgraph : Hnsw.Ba.Hgraph.t) (target : Lacaml.S.vec) ~num_neighbours_search ~num_neighbours :
  int Hnsw_algo.value_distance list =
  knn_on ~semantics:(if nearest_k_compat then 2 else 1) (index_of_ba hgraph) target
    ~ef:num_neighbours_search ~k:num_neighbours ~fill:1
  |> List.map (fun (node, distance_to_target) -> { Hnsw_algo.node; distance_to_target })

(* ---- the other operators of the header ----------------------------------------------------------- *)

(* distances.(q).(j) = distance (query q) (value ids.(q).(j)): EuclideanBa.distance / Ohnsw.distance_l2 gathered on the
   device (lib/hnsw.ml:809-815, lib/ohnsw.ml:899; bench_dist/bench_dist.ml:22-33 times exactly this, one pair per call) *)
let distance_batch (t : t) (queries : Lacaml.S.mat) (ids : int array array) : float array array =
  let nq = A2.dim2 queries in
  if Array.length ids <> nq then invalid_arg "distance_batch: one id list per query";
  let m = if nq = 0 then 0 else Array.length ids.(0) in
  let flat_ids = CArray.make int32_t (max 1 (nq * m)) and out = CArray.make float (max 1 (nq * m)) in
  Array.iteri (fun q row ->
      if Array.length row <> m then invalid_arg "distance_batch: ragged id lists";
      Array.iteri (fun j v -> CArray.set flat_ids (q * m + j) (Int32.of_int v)) row) ids;
  check (hnsw_distance_batch t.handle (bigarray_start array2 queries) (Int64.of_int nq) (Int64.of_int t.dim)
           (CArray.start flat_ids) (Int32.of_int m) (CArray.start out));
  Array.init nq (fun q -> Array.init m (fun j -> CArray.get out (q * m + j)))

(* Ohnsw.select_neighbours (lib/ohnsw.ml:647-663) for ONE base value: candidates as node ids (their distances to the
   target are recomputed on the device), at most ~num_neighbours kept, in selection order (nearest first).
   ~keep_all_if_few:true adds Hnsw_algo.SelectNeighbours' shortcut (lib/hnsw_algo.ml:596-599); ~degrees gives
   ~do_not_isolate:true (candidates of degree <= 1 are kept unconditionally, :591-592). *)
let select_neighbours ?(keep_all_if_few = false) ?degrees (t : t) (target : Lacaml.S.vec) ~(candidates : int list) ~num_neighbours : int list =
  let nc = List.length candidates in
  let cand = CArray.of_list int32_t (List.map Int32.of_int candidates) in
  let cnt = allocate int32_t (Int32.of_int nc) in
  let deg = match degrees with
    | Some l -> CArray.start (CArray.of_list int32_t (List.map Int32.of_int l))
    | None -> from_voidp int32_t null in
  let out = CArray.make int32_t (max 1 num_neighbours) and out_cnt = allocate int32_t 0l in
  check (hnsw_select_neighbours_batch t.handle (bigarray_start array1 target) 1L (Int64.of_int t.dim)
           (CArray.start cand) cnt (Int32.of_int (max 1 nc)) (Int32.of_int num_neighbours)
           (if keep_all_if_few then 1l else 0l) deg (CArray.start out) out_cnt);
  List.init (Int32.to_int !@out_cnt) (fun i -> Int32.to_int (CArray.get out i))

(* the device builder: the body of Ohnsw.build_batch_bigarray (lib/ohnsw.ml:840-857) in batches on the GPU
   (~max_batch:1 = Ohnsw.insert link for link); the OCaml builder stays the reference path, this is the fast one *)
let build ?(device = 0) ?(metric = 0) ?(seed = 0) ?(max_batch = 0) ?(batch_div = 0) ?(expected_ef = 0) ?(expected_semantics = 0) ~id_base ~num_connections
    ~num_nodes_search_construction (vectors : Lacaml.S.mat) : t =
  let dim = A2.dim1 vectors and n = A2.dim2 vectors in
  let b = make build_params in
  setf b b_num_connections (Int32.of_int num_connections); setf b b_efc (Int32.of_int num_nodes_search_construction);
  setf b b_metric (Int32.of_int metric); setf b b_id_base (Int32.of_int id_base);
  setf b b_seed (Unsigned.UInt64.of_int seed); setf b b_max_batch (Int32.of_int max_batch);
  setf b b_batch_div (Int32.of_int batch_div);
  setf b b_expected_ef (Int32.of_int expected_ef); setf b b_expected_semantics (Int32.of_int expected_semantics);
  let out = allocate index null in
  check (hnsw_build (bigarray_start array2 vectors) (Int64.of_int n) (Int32.of_int dim) (Int64.of_int dim) (addr b)
           (Int32.of_int device) out);
  let t = { handle = !@out; k_base = id_base; dim; scratch = Hashtbl.create 4 } in
  Gc.finalise (fun t -> ignore (hnsw_index_destroy t.handle)) t;
  t

(* flattened-index file (the reference has no persistence: lib/hnsw.ml:348, lib/ohnsw.ml:312 derive sexp with opaque values) *)
let save (t : t) (path : string) = check (hnsw_index_save t.handle path)
let load ?(device = 0) ~id_base ~dim (path : string) : t =
  let out = allocate index null in
  check (hnsw_index_load path (Int32.of_int device) out);
  let t = { handle = !@out; k_base = id_base; dim; scratch = Hashtbl.create 4 } in
  Gc.finalise (fun t -> ignore (hnsw_index_destroy t.handle)) t;
  t

(* Hgraph.Stats (lib/hnsw.ml:353-375) of one layer: (layer size, min, max, mean degree, number of isolated nodes) *)
let stats (t : t) ~layer : int * int * int * float * int =
  let s = make layer_stats in
  check (hnsw_index_layer_stats t.handle (Int32.of_int layer) (addr s));
  (Int64.to_int (getf s ls_num_nodes), Int32.to_int (getf s ls_min_degree), Int32.to_int (getf s ls_max_degree),
   getf s ls_mean_degree, Int64.to_int (getf s ls_num_isolated))

(* mima.isolated of one layer (lib/hnsw.ml:357,364-366): the nodes without a neighbour, in the reference's list order
   (descending: consed during the ascending Map.fold) *)
let isolated (t : t) ~layer : int list =
  let cnt = allocate int64_t 0L in
  check (hnsw_index_layer_isolated t.handle (Int32.of_int layer) (from_voidp int64_t null) 0L cnt);
  let c = Int64.to_int !@cnt in
  if c = 0 then [] else begin
    let ids = CArray.make int64_t c in
    check (hnsw_index_layer_isolated t.handle (Int32.of_int layer) (CArray.start ids) (Int64.of_int c) cnt);
    List.map Int64.to_int (CArray.to_list ids)
  end

(* the permutation of 0 .. n-1 behind the option "visited_blocks" (hnsw_index_locality_codes): introspection only.  The
   library writes the INDEX's n codes: the length comes from the handle, never from the caller *)
let locality_codes (t : t) : (int32, Bigarray.int32_elt, Bigarray.c_layout) A1.t =
  let inf = make index_info in
  check (hnsw_index_get_info t.handle (addr inf));
  let out = A1.create Bigarray.int32 Bigarray.c_layout (max 1 (Int64.to_int (getf inf ii_n))) in
  check (hnsw_index_locality_codes t.handle (bigarray_start array1 out));
  out

(* everything the first search with this ef (and accept rule) would do once -- the visited-structure decision, the kernel's
   residency, its code object -- now (hnsw_index_prepare); [create ~expected_ef] / [build ~expected_ef] call it themselves *)
let prepare ?(semantics = 0) (t : t) ~ef : unit =
  let p = make search_params in
  setf p p_ef (Int32.of_int ef); setf p p_k 1l; setf p p_fill 0l; setf p p_semantics (Int32.of_int semantics);
  check (hnsw_index_prepare t.handle (addr p))

(* 0: searches at this ef use the tag cache; else log2 of the bitmap-block slots (hnsw_index_visited_blocks) *)
let visited_blocks ?(semantics = 0) (t : t) ~ef : int =
  let p = make search_params in
  setf p p_ef (Int32.of_int ef); setf p p_k 1l; setf p p_fill 0l; setf p p_semantics (Int32.of_int semantics);
  let out = allocate int32_t 0l in
  check (hnsw_index_visited_blocks t.handle (addr p) out);
  Int32.to_int !@out

(* Hgraph.Stats.compute (lib/hnsw.ml:370-375) as the reference's record: per layer (size, {min; max; mean; isolated}) *)
let stats_compute (t : t) ~max_layer : (int * (int * int * float * int list)) list =
  List.init (max_layer + 1) (fun layer ->
      let (size, mi, ma, mean, _) = stats t ~layer in
      (size, (mi, ma, mean, isolated t ~layer)))

(* Page-lock a query or result matrix a benchmark loop passes again and again (benchmark/benchmark.ml:86-98): the
   copies of knn_batch* then run at PCIe speed.  The CALLER keeps the Bigarray reachable until [unpin] -- the library
   never registers memory on its own. *)
let pin (m : (_, _, _) A2.t) = check (hnsw_host_register (to_voidp (bigarray_start array2 m)) (Int64.of_int (A2.size_in_bytes m)))
let unpin (m : (_, _, _) A2.t) = check (hnsw_host_unregister (to_voidp (bigarray_start array2 m)))

(* the graph of a device index as a [flat] (the input of unflatten_ohnsw / unflatten_ba): e.g. an index built on the
   device with [build], handed back to the OCaml builder so that Ohnsw.insert can go on from there *)
let export (t : t) : flat =
  let inf = make index_info in
  check (hnsw_index_get_info t.handle (addr inf));
  let n = Int64.to_int (getf inf ii_n) and max_layer = Int32.to_int (getf inf ii_max_layer) in
  let width0 = Int32.to_int (getf inf ii_max_degree0) and width_upper = max 1 (Int32.to_int (getf inf ii_max_degree)) in
  let deg0 = A1.create Bigarray.int32 Bigarray.c_layout n in
  let nbr0 = A2.create Bigarray.int32 Bigarray.c_layout n width0 in
  check (hnsw_index_export_layer0 t.handle (bigarray_start array1 deg0) (bigarray_start array2 nbr0));
  let upper = Array.init max_layer (fun l ->
      let cnt = allocate int64_t 0L in
      check (hnsw_index_export_upper_count t.handle (Int32.of_int (l + 1)) cnt);
      let c = Int64.to_int !@cnt in
      let nodes = A1.create Bigarray.int64 Bigarray.c_layout c in
      let deg = A1.create Bigarray.int32 Bigarray.c_layout c in
      let nbr = A2.create Bigarray.int32 Bigarray.c_layout c width_upper in
      if c > 0 then
        check (hnsw_index_export_upper t.handle (Int32.of_int (l + 1)) (bigarray_start array1 nodes)
                 (bigarray_start array1 deg) (bigarray_start array2 nbr));
      (nodes, deg, nbr)) in
  { deg0; nbr0; upper; max_layer; width0; width_upper; entry_point = Int64.to_int (getf inf ii_entry_point) }
